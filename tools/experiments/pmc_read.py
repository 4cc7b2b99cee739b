"""Per-kernel PMC sums of the last dispatches from rocprofv3 sqlite outputs: python tools/experiments/pmc_read.py gpurun_out/pmc_<tag>_* [kernel substring ...]"""
import sqlite3, sys, glob, os
dirs = [a for a in sys.argv[1:] if os.path.isdir(a)]
pats = [a for a in sys.argv[1:] if not os.path.isdir(a)] or ["delta_c1_", "delta_c2_", "delta_prepare_split", "c3_dense"]
for d in dirs:
    for dbf in glob.glob(os.path.join(d, "*_results.db")):
        db = sqlite3.connect(dbf); cur = db.cursor()
        tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
        pmc = [t for t in tabs if 'pmc_event' in t][0]; disp = [t for t in tabs if 'kernel_dispatch' in t][0]
        info = [t for t in tabs if 'info_pmc' in t][0]; sym = [t for t in tabs if 'kernel_symbol' in t][0]
        for pat in pats:
            q = f"""select d.id, d.start, d.end, i.name, sum(p.value) from {pmc} p join {disp} d on p.event_id = d.event_id
                    join {info} i on p.pmc_id = i.id join {sym} s on d.kernel_id = s.id where s.kernel_name like '%{pat}%' group by d.id, i.name"""
            byd = {}
            for did, st, en, cn, val in cur.execute(q):
                byd.setdefault(did, {"dur_us": (en - st) / 1e3})[cn] = val
            ids = sorted(byd)[-5:]
            if not ids: continue
            keys = [k for k in byd[ids[0]] if k != "dur_us"]
            avg = {k: sum(byd[i][k] for i in ids) / len(ids) for k in ["dur_us"] + keys}
            print(os.path.basename(d), pat, " ".join(f"{k}={avg[k]:.4g}" for k in avg))
