"""Parity AT SCALE on the benchmarked path: BASELINE.json configs[1] (1 query vs 1024 leg-computed candidates, C = 4)
exactly as bench.py runs it -- query leg + candidate legs from the seeded pool images, cached spectra, Delta head + spectral
correlation head -- with ALL 1024 (overlap, yaw) results compared with the fp64 oracle, for every arithmetic mode and for
two weight sets (Glorot-scaled and trained-like dynamic range).

The oracle side is tests/golden/parity_sweep_<set>.npz, produced by tests/golden/make_parity_sweep_golden.py (the fp64
oracle needs ~3 min per 1024 pairs on 8 cores; the GPU box's minutes are better spent on the GPU).  The CPU test below
re-runs the oracle on a few pairs and requires equality with the file, so the file cannot drift from the recipe.

Gates (north star): |d overlap| <= 1e-4 on every pair; yaw bin identical except where the ORACLE's own top-2 gap is
below 1e-5 relative (listed in the report); |d logit| <= 1e-3 (1 + |logit|).
The statistics go to gpurun_out/parity/<case>.json, one file per case; tools/collect_parity.py merges them into the tracked
profiles/r<round>_parity.json."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import overlapnet_oracle as O
from tools import synthetic as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = S.REFERENCE_MODEL_CFG
# (weight set, channels, pool): the benchmark configuration with both weight sets, and a shorter sweep at the other two channel
# counts the reference's configs produce (depth only; depth + normals + intensity) with the wide-range weights
# the 4096-candidate pool crosses the 2048-pair chunk boundary of the heads and the 1024-scan slices of the leg
CASES = [("glorot", 4, 1024), ("trained_like", 4, 1024), ("trained_like", 1, 128), ("trained_like", 5, 128), ("trained_like", 4, 4096)]

# (name, leg arithmetic, head arithmetic, correlation form); the first row is what bench.py and `Infer` run by default
# "default" prepares every pair in scratch; "default_cached" is the route bench.py and `Infer` take: the candidates' Delta cache rows
# (ovn_delta_cache) + the query's shared block, with the dead-channel compaction of the query (both routes vs the oracle)
MODES = [("default", None, None, "spectral"),
         ("default_cached", None, None, "spectral_cached"),
         ("all_f32", "f32", "f32", "direct")]


def _golden(name, C, POOL):
    suffix = "" if (POOL, C) == (1024, 4) else "_c%d_p%d" % (C, POOL)
    with np.load(os.path.join(ROOT, "tests", "golden", "parity_sweep_%s%s.npz" % (name, suffix))) as z:
        return {k: z[k] for k in z.files}


@pytest.mark.parametrize("wset,C,POOL", CASES)
def test_golden_file_equals_live_oracle_on_sample(wset, C, POOL):
    """CPU: the committed oracle outputs are what the oracle gives today on inputs rebuilt from the seeds."""
    g = _golden(wset, C, POOL)
    assert int(g["pool"][0]) == POOL and g["overlap"].shape == (POOL,)
    w = S.WEIGHT_SETS[wset](C)
    fx = S.load_fixture_images()
    qfv = O.leg_forward(S.sweep_query_image(C, fx), w, CFG, np.float64)
    assert abs(qfv.sum() - g["query_feat_sum"][0]) <= 1e-9 * abs(g["query_feat_sum"][0])
    s, imgs = next(S.sweep_pool_images(POOL, C, 0, fx))
    idx = np.array([0, 5, 77])
    fv = O.leg_forward(imgs[idx], w, CFG, np.float64)
    ov, yaw, lg, corr = O.heads_forward(fv, np.repeat(qfv, len(idx), axis=0), w)
    np.testing.assert_allclose(ov, g["overlap"][idx], rtol=0, atol=1e-12)
    np.testing.assert_allclose(lg, g["logit"][idx], rtol=1e-10, atol=1e-10)
    assert np.array_equal(yaw, g["yaw"][idx])
    # the golden logits are spread (a saturated or constant set would make the overlap gate vacuous)
    assert g["logit"].max() - g["logit"].min() > (10.0 if wset == "trained_like" else 2.0)


def _stats(d):
    d = np.abs(np.asarray(d, np.float64))
    return {"max": float(d.max()), "p99": float(np.percentile(d, 99)), "mean": float(d.mean())}


@pytest.mark.gpu
@pytest.mark.skipif(not torch.cuda.is_available(), reason="needs an MI355X")
@pytest.mark.parametrize("wset,C,POOL", CASES)
def test_sweep_every_pair_against_oracle(wset, C, POOL):
    from overlapnet_amd.engine import OvnEngine
    g = _golden(wset, C, POOL)
    w = S.WEIGHT_SETS[wset](C)
    eng = OvnEngine(64, 900, C)
    eng.load_weights(w, CFG)
    fx = S.load_fixture_images()
    dev = eng.device
    pool_imgs = [(s, torch.from_numpy(imgs).to(dev)) for s, imgs in S.sweep_pool_images(POOL, C, 0, fx)]
    qimg = torch.from_numpy(S.sweep_query_image(C, fx)).to(dev)
    report = {"workload": "1-vs-%d sweep, 64x900x%d, weight set '%s': images -> leg -> spectra -> Delta + correlation heads "
                          "on the GPU vs the fp64 oracle on the same images, every pair compared" % (POOL, C, wset),
              "oracle_logit_range": [float(g["logit"].min()), float(g["logit"].max())], "modes": {}}
    default_leg, default_head = eng.leg_precision, eng.head_precision
    failures = []
    try:
        for name, leg_p, head_p, corr_form in MODES:
            eng.set_leg_precision(leg_p or default_leg)
            eng.set_head_precision(head_p or default_head)
            cands = torch.empty((POOL, 360, 128), dtype=torch.float32, device=dev)
            for s, timg in pool_imgs:
                eng.leg(timg, out=cands[s:s + timg.shape[0]])
            qfv = eng.leg(qimg)
            if corr_form == "spectral":
                r = eng.heads(cands, qfv, spec_l=eng.spectrum(cands), spec_r=eng.spectrum(qfv), want_logit=True)
            elif corr_form == "spectral_cached":
                r = eng.heads(cands, qfv, spec_l=eng.spectrum(cands), spec_r=eng.spectrum(qfv), want_logit=True,
                              dcache_l=eng.delta_cache(cands))
            else:
                r = eng.heads(cands, qfv, want_logit=True)
            torch.cuda.synchronize()
            ov, lg, yaw = r["overlap"].cpu().numpy(), r["logit"].cpu().numpy(), r["yaw"].cpu().numpy()
            d_ov, d_lg = ov - g["overlap"], lg - g["logit"]
            bad = np.nonzero(yaw != g["yaw"])[0]
            rec = {"leg": eng.leg_precision, "head": eng.head_precision, "corr": corr_form, "pairs": POOL,
                   "k_walk": eng.head_walk_stats() if head_p is None else None,
                   "abs_d_overlap": _stats(d_ov), "abs_d_logit": _stats(d_lg),
                   "feature_sum_rel_err_max": float(np.max(np.abs(cands.double().sum(dim=(1, 2)).cpu().numpy() - g["feat_sum"])
                                                           / np.abs(g["feat_sum"]))),
                   "yaw_mismatches": [{"pair": int(i), "gpu": int(yaw[i]), "oracle": int(g["yaw"][i]),
                                       "oracle_top2_gap_rel": float(g["corr_top2_gap"][i])} for i in bad]}
            report["modes"][name] = rec
            print("[%s/%s] max |d overlap| %.3g  p99 %.3g  max |d logit| %.3g  yaw mismatches %d" %
                  (wset, name, rec["abs_d_overlap"]["max"], rec["abs_d_overlap"]["p99"], rec["abs_d_logit"]["max"], len(bad)))
            if rec["abs_d_overlap"]["max"] > 1e-4:
                failures.append("%s: max |d overlap| %.3g" % (name, rec["abs_d_overlap"]["max"]))
            if np.any(np.abs(d_lg) > 1e-3 * (1 + np.abs(g["logit"]))):
                failures.append("%s: logit error %.3g" % (name, rec["abs_d_logit"]["max"]))
            hard = [m for m in rec["yaw_mismatches"] if m["oracle_top2_gap_rel"] > 1e-5]
            if hard:
                failures.append("%s: yaw bins differ on pairs with a clear maximum: %s" % (name, hard))
    finally:
        eng.close()
    # one report file per case (a fresh GPU box starts with an empty gpurun_out/, and partial runs must not overwrite each other);
    # tools/collect_parity.py merges them into the tracked profiles/r<round>_parity.json
    out_dir = os.path.join(ROOT, "gpurun_out", "parity")
    os.makedirs(out_dir, exist_ok=True)
    case = wset if (C, POOL) == (4, 1024) else "%s_c%d_p%d" % (wset, C, POOL)
    json.dump(report, open(os.path.join(out_dir, case + ".json"), "w"), indent=1)
    assert not failures, failures
    # the default arithmetic has the error of an fp32 evaluation: within 2x of the all-fp32 mode (+ 2e-6 of fp32 noise floor)
    m = report["modes"]
    for name in ("default", "default_cached"):
        assert m[name]["abs_d_overlap"]["max"] <= 2 * m["all_f32"]["abs_d_overlap"]["max"] + 2e-6, \
            (name, m[name]["abs_d_overlap"], m["all_f32"]["abs_d_overlap"])


# ---- the full stack: raw clouds -> projection + normals -> leg -> heads (BASELINE configs[4] as bench.py's `fullstack` step runs it) ----
FULLSTACK = [("glorot", "parity_fullstack.npz"), ("trained_like", "parity_fullstack_trained_like.npz")]


@pytest.mark.parametrize("wset,fname", FULLSTACK)
def test_fullstack_golden_equals_live_oracle_on_sample(wset, fname):
    """CPU: the committed fp64-oracle outputs of the fullstack step (tests/golden/make_fullstack_golden.py) are what the oracle gives
    today from the same raw clouds (its own projection with the restated NumPy float32 angles, normals, fp64 leg and heads)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_fullstack_golden", os.path.join(ROOT, "tests", "golden", "make_fullstack_golden.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    with np.load(os.path.join(ROOT, "tests", "golden", fname)) as z:
        g = {k: z[k] for k in z.files}
    assert int(g["query_cloud"][0]) == m.QUERY_CLOUD and len(g["overlap"]) == 64
    ids = [0, 37]
    ov, yaw, lg = m.oracle_fullstack(ids, weights=S.WEIGHT_SETS[wset](4))
    np.testing.assert_allclose(ov, g["overlap"][ids], rtol=0, atol=1e-12)
    assert np.array_equal(yaw, g["yaw"][ids])
    # the clouds are rotations of the two scans about z: the yaw the network reports follows the rotation (37 columns = 14.8 degrees)
    assert np.all(np.diff(g["yaw"][:8]) >= 14) and np.all(np.diff(g["yaw"][:8]) <= 19)


@pytest.mark.gpu
@pytest.mark.parametrize("wset,fname", FULLSTACK)
def test_fullstack_from_raw_clouds_against_the_committed_oracle(wset, fname):
    """GPU: 64 candidates + the query from RAW clouds through `ovn_project` (stacked leg input), the leg and both heads, against the
    fp64 oracle that started from the same clouds -- the check bench.py makes on its timed `fullstack` step, as a test, for both
    weight sets and both arithmetic modes."""
    from overlapnet_amd.engine import OvnEngine
    from overlapnet_amd import preprocess as P
    with np.load(os.path.join(ROOT, "tests", "golden", fname)) as z:
        g = {k: z[k] for k in z.files}
    fx = S.load_fixture_images()
    ids = list(range(64)) + [int(g["query_cloud"][0])]
    e = OvnEngine(64, 900, 4)
    w = S.WEIGHT_SETS[wset](4)
    e.load_weights(w, CFG)
    try:
        r = P.project_scans([S.fullstack_cloud(fx, i) for i in ids], engine=e, want=(), stacked_flags=(True, True, False))
        for leg_mode, head_mode, corr in (("f16x3", "f16x3", "spectral"), ("f32", "f32", "direct")):
            e.set_leg_precision(leg_mode)
            e.set_head_precision(head_mode)
            fv = e.leg(r["stacked"])
            if corr == "spectral":
                sp = e.spectrum(fv)
                out = e.heads(fv[:64].contiguous(), fv[64:65].contiguous(), spec_l=sp[:64].contiguous(), spec_r=sp[64:65].contiguous(), want_logit=True)
            else:
                out = e.heads(fv[:64].contiguous(), fv[64:65].contiguous(), want_logit=True)
            ov, yaw, lg = out["overlap"].cpu().numpy(), out["yaw"].cpu().numpy(), out["logit"].cpu().numpy()
            assert np.max(np.abs(ov - g["overlap"])) <= 1e-4, (leg_mode, float(np.max(np.abs(ov - g["overlap"]))))
            assert np.array_equal(yaw, g["yaw"])
            assert np.max(np.abs(lg - g["logit"]) / (1 + np.abs(g["logit"]))) <= 1e-3
    finally:
        e.close()
