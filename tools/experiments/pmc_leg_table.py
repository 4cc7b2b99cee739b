"""Per-kernel PMC table of the batched leg from the three passes of tools/experiments/pmc_leg.sh:
    python tools/experiments/pmc_leg_table.py gpurun_out/pmc_<tag>_1 gpurun_out/pmc_<tag>_2 gpurun_out/pmc_<tag>_3
Only the 256-scan dispatches of a kernel are averaged (duration > half of the kernel's longest dispatch)."""
import glob, os, sqlite3, sys

KERNELS = ["leg_front", "conv_strip2_kernelILi32E", "conv_strip2_kernelILi64ELi3E", "conv_strip2_kernelILi64ELi2E", "leg_tail"]   # mangled names: front, s_conv3, s_conv3a, s_conv4 (batched: conv_strip2_kernel), tail
agg = {}
for d in sys.argv[1:]:
    for dbf in glob.glob(os.path.join(d, "*_results.db")):
        cur = sqlite3.connect(dbf).cursor()
        tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
        pmc = [t for t in tabs if "pmc_event" in t][0]
        disp = [t for t in tabs if "kernel_dispatch" in t][0]
        info = [t for t in tabs if "info_pmc" in t][0]
        sym = [t for t in tabs if "kernel_symbol" in t][0]
        for pat in KERNELS:
            q = ("select d.id, d.start, d.end, i.name, sum(p.value) from %s p join %s d on p.event_id = d.event_id join %s i on "
                 "p.pmc_id = i.id join %s s on d.kernel_id = s.id where s.kernel_name like '%%%s%%' group by d.id, i.name" % (pmc, disp, info, sym, pat))
            byd = {}
            for did, st, en, cn, val in cur.execute(q):
                byd.setdefault(did, {"dur_us": (en - st) / 1e3})[cn] = val
            if not byd:
                continue
            top = max(v["dur_us"] for v in byd.values())
            big = [v for v in byd.values() if v["dur_us"] > 0.5 * top]
            for k in big[0]:
                agg.setdefault(pat, {}).setdefault(k if k != "dur_us" else "dur_us_" + os.path.basename(d), sum(v[k] for v in big) / len(big))
print("| kernel | us | GHz | matrix pipe busy | issuing / issue-stalled / parked | LDS conflict / LDS active | MFMA insts |")
print("|---|---|---|---|---|---|---|")
for pat, a in agg.items():
    dur = [v for k, v in a.items() if k.startswith("dur_us_")]
    g = a.get("GRBM_GUI_ACTIVE", 0) / 8
    d1 = [v for k, v in a.items() if k.startswith("dur_us_") and k.endswith("_1")]
    ghz = g / (d1[0] * 1e3) if d1 and g else 0
    busy = a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / g if g else 0
    wc = a.get("SQ_WAVE_CYCLES", 0) or 1
    print("| %s | %.0f | %.2f | %.0f %% | %.0f / %.0f / %.0f %% | %.0f %% | %.3g |" % (
        pat, sum(dur) / len(dur), ghz, 100 * busy, 100 * a.get("SQ_ACTIVE_INST_ANY", 0) / wc, 100 * a.get("SQ_WAIT_INST_ANY", 0) / wc,
        100 * a.get("SQ_WAIT_ANY", 0) / wc, 100 * a.get("SQ_LDS_BANK_CONFLICT", 0) / max(a.get("SQ_ACTIVE_INST_LDS", 1), 1), a.get("SQ_INSTS_MFMA", 0)))
