"""Budget map of the north star's tolerance (|d overlap| <= 1e-4, exact yaw bin): how much of it does each arithmetic mode spend
when the network's dynamic range moves away from the benchmark's -- trained-like weights with the feature volumes scaled by 0.5 / 2 /
4, depth-only images at the 50 m maximum range, and scans without a single valid pixel (VERDICT r4 "next" item 7).

Oracle side: tests/golden/parity_budget.npz (fp64, tests/golden/make_parity_budget_golden.py; the CPU test re-runs a sample).  The
GPU test writes gpurun_out/parity/budget.json; the round's copy is profiles/r5_parity_budget.json.  Gates: the north star's, on every
case, for BOTH modes -- a case that leaves the budget is reported with its fp32-mode twin, and fails."""
import importlib.util
import json
import os

import numpy as np
import pytest

from tools import synthetic as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = S.REFERENCE_MODEL_CFG
_spec = importlib.util.spec_from_file_location("make_parity_budget_golden", os.path.join(ROOT, "tests", "golden", "make_parity_budget_golden.py"))
B = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(B)


def _golden():
    with np.load(os.path.join(ROOT, "tests", "golden", "parity_budget.npz")) as z:
        return {k: z[k] for k in z.files}


@pytest.mark.parametrize("name", ["gain_2", "invalid"])
def test_budget_golden_equals_live_oracle_on_sample(name):
    from oracle import overlapnet_oracle as O
    g = _golden()
    C, w, cands, quers = B.budget_case(name)
    idx = [0, 3] if name != "invalid" else [0, 1, 3]
    fv = O.leg_forward(cands[idx], w, CFG, np.float64)
    qf = O.leg_forward(quers[idx] if quers.shape[0] > 1 else quers, w, CFG, np.float64)
    if qf.shape[0] == 1:
        qf = np.repeat(qf, len(idx), axis=0)
    ov, yaw, lg, _ = O.heads_forward(fv, qf, w)
    np.testing.assert_allclose(ov, g[name + "/overlap"][idx], rtol=0, atol=1e-12)
    assert np.array_equal(yaw, g[name + "/yaw"][idx])
    if name == "gain_2":     # the last-layer gain scales the feature volumes exactly
        fv1 = O.leg_forward(cands[idx], S.make_trained_like_weights(4), CFG, np.float64)
        np.testing.assert_allclose(fv, 2.0 * fv1, rtol=1e-6, atol=1e-9)      # (weights are rounded to float32 after the scaling)


@pytest.mark.gpu
def test_tolerance_budget_map():
    import torch
    from overlapnet_amd.engine import OvnEngine
    g = _golden()
    report, failures = {}, []
    for name in B.CASE_NAMES:
        C, w, cands, quers = B.budget_case(name)
        eng = OvnEngine(64, 900, C)
        eng.load_weights(w, CFG)
        dev = eng.device
        n = cands.shape[0]
        per_pair_q = quers.shape[0] > 1
        rec = {"pairs": int(n), "channels": C, "oracle_logit_range": [float(g[name + "/logit"].min()), float(g[name + "/logit"].max())],
               "oracle_feature_max": float(g[name + "/feat_max"].max()),
               "oracle_near_tie_pairs": int((g[name + "/corr_top2_gap"] < 1e-5).sum()), "modes": {}}
        try:
            for mode, leg_p, head_p, corr in (("default", "f16x3", "f16x3", "spectral"), ("all_f32", "f32", "f32", "direct")):
                eng.set_leg_precision(leg_p)
                eng.set_head_precision(head_p)
                fv = eng.leg(torch.from_numpy(cands).to(dev))
                qf = eng.leg(torch.from_numpy(quers).to(dev))
                assert bool(torch.isfinite(fv).all()) and bool(torch.isfinite(qf).all())
                kw = {}
                if per_pair_q:
                    kw = dict(lidx=np.arange(n, dtype=np.int32), ridx=np.arange(n, dtype=np.int32))
                if corr == "spectral":
                    r = eng.heads(fv, qf, spec_l=eng.spectrum(fv), spec_r=eng.spectrum(qf), want_logit=True, **kw)
                else:
                    r = eng.heads(fv, qf, want_logit=True, **kw)
                ov, lg, yaw = r["overlap"].cpu().numpy(), r["logit"].cpu().numpy(), r["yaw"].cpu().numpy()
                d_ov = np.abs(ov - g[name + "/overlap"])
                d_lg = np.abs(lg - g[name + "/logit"])
                bad = np.nonzero(yaw != g[name + "/yaw"])[0]
                hard = [int(i) for i in bad if g[name + "/corr_top2_gap"][i] > 1e-5]
                rec["modes"][mode] = {"max_abs_d_overlap": float(d_ov.max()), "budget_used": float(d_ov.max() / 1e-4),
                                      "max_abs_d_logit": float(d_lg.max()),
                                      "max_rel_d_logit": float(np.max(d_lg / (1 + np.abs(g[name + "/logit"])))),
                                      "yaw_mismatches": int(len(bad)), "yaw_mismatches_with_a_clear_oracle_maximum": len(hard)}
                print("[budget %-9s %-8s] |d overlap| %.3g (%.0f %% of 1e-4)  |d logit| %.3g  yaw mismatches %d (%d clear)"
                      % (name, mode, d_ov.max(), 100 * d_ov.max() / 1e-4, d_lg.max(), len(bad), len(hard)))
                if d_ov.max() > 1e-4:
                    failures.append("%s/%s: max |d overlap| %.3g" % (name, mode, d_ov.max()))
                if hard:
                    failures.append("%s/%s: yaw bins differ where the oracle's maximum is clear: pairs %s" % (name, mode, hard))
        finally:
            eng.close()
        report[name] = rec
    out_dir = os.path.join(ROOT, "gpurun_out", "parity")
    os.makedirs(out_dir, exist_ok=True)
    json.dump(report, open(os.path.join(out_dir, "budget.json"), "w"), indent=1)
    assert not failures, failures
