#!/usr/bin/env python3
"""Print average PMC counter values per dispatch for kernels matching a substring.
   python tools/pmc_report.py gpurun_out/pmc_x delta"""
import glob, os, sqlite3, sys
src, pat = sys.argv[1], sys.argv[2]
for db in sorted(glob.glob(os.path.join(src, "**", "*.db"), recursive=True)):
    c = sqlite3.connect(db)
    try:
        rows = c.execute("select kernel_name, counter_name, avg(value), count(*), avg(duration) from counters_collection "
                         "where kernel_name like ? group by kernel_name, counter_name", ("%" + pat + "%",)).fetchall()
    except Exception as e:
        print(db, "ERR", e); continue
    for k, cn, v, n, d in rows:
        print("%-28s %-34s %14.5g  n=%d  dur_us=%.1f" % (k.split("(")[0][-28:], cn, v, n, d / 1e3))
