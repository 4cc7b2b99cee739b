#!/usr/bin/env python3
"""Summarise the rocprofv3 (rocpd sqlite) outputs of tools/profile_bench.sh into profiles/<tag>_*.{md,json}.

    python tools/summarize_rocprof.py gpurun_out/prof_r1 r1

kernel-trace pass  -> per-kernel calls / total / average duration (the `top_kernels` view = `--stats`)
PMC passes         -> per-kernel average counter values per dispatch.
HBM bytes follow MI355X_MICROARCH.md section HBM: FETCH_SIZE/WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE
counts 128-B requests as 64 B for wide coalesced streams, so the read side is DOUBLED before it is compared with a
byte count (WRITE_SIZE is taken as is, uncalibrated).
"""
import json
import os
import sqlite3
import sys


def short(name):
    """Kernel name without namespace / argument list; rocprofv3 leaves some template kernels mangled
    (_ZN12_GLOBAL__N_1<len><name>I...E...): the <len> characters after the length prefix are the plain name."""
    import re
    m = re.match(r"_ZN\d+_GLOBAL__N_1(\d+)", name)
    if m:
        k = int(m.group(1))
        return name[m.end():m.end() + k]
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0]


def main():
    src, tag = sys.argv[1], sys.argv[2]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dst = os.path.join(root, "profiles")
    os.makedirs(dst, exist_ok=True)
    out = {"tag": tag, "kernels": [], "counters": {}}
    con = sqlite3.connect(os.path.join(src, "stats", "bench_results.db"))
    rows = con.execute("select name,total_calls,total_duration,average,percentage from top_kernels").fetchall()
    lines = ["# rocprofv3 --kernel-trace --stats summary (%s)" % tag, "",
             "command: `python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras` (durations in us)", "",
             "| kernel | calls | total us | avg us | % |", "|---|---|---|---|---|"]
    for name, calls, tot, avg, pct in rows:
        out["kernels"].append({"name": short(name), "calls": calls, "total_us": tot, "avg_us": avg, "pct": pct})
        lines.append("| `%s` | %d | %.1f | %.2f | %.2f |" % (short(name), calls, tot, avg, pct))
    # the timed launches only: bench.py ran 2 warm-up + 5 timed steps (+ 2 untimed sweeps of query 0 afterwards, without the next
    # query's leg on the side stream), so dispatches 2 .. 6 of the head kernels are the ones its own HIP-event timer brackets (the
    # first launches run on cold caches and pull the plain average up)
    lines += ["", "| kernel | avg us over the 5 timed dispatches (2 warm-up launches before, 2 untimed after) | bench.py event timer in this profiled run |", "|---|---|---|"]
    ev = None
    logp = os.path.join(src, "stats.log")
    if os.path.isfile(logp):
        import re
        m = re.search(r'"avg_launch_ms": ([0-9.]+)', open(logp, errors="ignore").read())
        ev = float(m.group(1)) * 1e3 if m else None
    out["timed_launches"] = {}
    for (name,) in con.execute("select distinct name from kernels where name like '%delta_c1%' or name like '%delta_c2_%' or name like '%delta_prepare%' or name like '%c3_dense%'").fetchall():
        d = [r[0] for r in con.execute("select duration from kernels where name = ? order by start", (name,)).fetchall()]
        last = d[2:7] if len(d) >= 7 else d[-5:]
        avg = sum(last) / len(last) / 1e3
        out["timed_launches"][short(name)] = {"avg_us_last5": avg, "dispatches": len(d)}
        lines.append("| `%s` | %.2f | %s |" % (short(name), avg, ("%.2f" % ev) if (ev and ("delta_c12" in name or "delta_c1_" in name)) else "-"))
    # per-dispatch geometry of our kernels
    geo = con.execute("select name, max(grid_x), max(workgroup_x), max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), "
                      "max(lds_size) from kernels group by name").fetchall()
    lines += ["", "| kernel | grid_x (max) | wg | vgpr | agpr | sgpr | lds B |", "|---|---|---|---|---|---|---|"]
    for g in geo:
        lines.append("| `%s` | %s | %s | %s | %s | %s | %s |" % ((short(g[0]),) + tuple(g[1:])))
    for sub in ("pmc_fetch", "pmc_write", "pmc_mfma", "pmc_clk"):
        p = os.path.join(src, sub, "bench_results.db")
        if not os.path.isfile(p):
            continue
        c = sqlite3.connect(p)
        q = ("select kernel_name, counter_name, avg(value), count(*), avg(duration) from counters_collection "
             "group by kernel_name, counter_name")
        for kname, cname, val, cnt, dur in c.execute(q):
            out["counters"].setdefault(short(kname), {})[cname] = {"avg_per_dispatch": val, "dispatches": cnt,
                                                                  "avg_duration_ns_profiled": dur}
    lines += ["", "## PMC counters (average per dispatch)", "", "| kernel | counter | avg/dispatch | dispatches |", "|---|---|---|---|"]
    for k, cs in sorted(out["counters"].items()):
        for cn, v in sorted(cs.items()):
            lines.append("| `%s` | %s | %.4g | %d |" % (k, cn, v["avg_per_dispatch"], v["dispatches"]))
    # HBM traffic of the head kernels per launch
    lines += ["", "## HBM traffic per launch (bytes) = 2 x FETCH_SIZE KiB (gfx950 correction) + WRITE_SIZE KiB", ""]
    traffic = {}
    for k, cs in out["counters"].items():
        if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
            rd = 2.0 * cs["FETCH_SIZE"]["avg_per_dispatch"] * 1024
            wr = cs["WRITE_SIZE"]["avg_per_dispatch"] * 1024
            traffic[k] = {"read_bytes": rd, "write_bytes": wr, "total_bytes": rd + wr}
            lines.append("* `%s`: read %.3e  write %.3e  total %.3e" % (k, rd, wr, rd + wr))
    out["hbm_traffic_per_launch"] = traffic
    open(os.path.join(dst, "%s_rocprof_summary.md" % tag), "w").write("\n".join(lines) + "\n")
    json.dump(out, open(os.path.join(dst, "%s_rocprof_summary.json" % tag), "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
