#!/usr/bin/env python3
"""Merge the per-case parity reports that tests/test_parity_sweep.py leaves under gpurun_out/parity/ into ONE tracked file:

    python tools/collect_parity.py profiles/r3_parity.json

Refuses to write a file that lacks any case of the test's CASES list (a partial GPU run must not replace a complete report)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from tests.test_parity_sweep import CASES
    dst = sys.argv[1]
    src = os.path.join(ROOT, "gpurun_out", "parity")
    want = [w if (c, p) == (4, 1024) else "%s_c%d_p%d" % (w, c, p) for w, c, p in CASES]
    missing = [k for k in want if not os.path.isfile(os.path.join(src, k + ".json"))]
    if missing:
        raise SystemExit("missing parity cases under %s: %s" % (src, missing))
    merged = {k: json.load(open(os.path.join(src, k + ".json"))) for k in want}
    json.dump(merged, open(dst, "w"), indent=1)
    for k in want:
        m = merged[k]["modes"]
        print("%-28s" % k, "  ".join("%s: max|d ov| %.2e yaw mism %d" % (n, r["abs_d_overlap"]["max"], len(r["yaw_mismatches"]))
                                     for n, r in m.items()))


if __name__ == "__main__":
    main()
