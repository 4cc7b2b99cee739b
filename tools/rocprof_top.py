#!/usr/bin/env python3
"""Per-kernel calls / average / total duration from a rocprofv3 (rocpd sqlite) kernel trace:  python tools/rocprof_top.py <results.db> [rows]"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
rows = con.execute("select name,total_calls,total_duration,average,percentage from top_kernels").fetchall()
for name, calls, tot, avg, pct in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    print("%-90s calls %5d  avg %9.3f ms  total %10.2f ms  %5.1f %%" % (name.replace("(anonymous namespace)::", "").replace("void ", "")[:90], calls, avg / 1e3, tot / 1e3, pct))
