"""GPU: two processes run the REAL kernels on their shards of one sweep (both on cuda:0, gloo rendezvous, host tensors in the
collectives -- the box has one GPU; on a multi-GPU node the same code runs one rank per GPU over RCCL) and must reproduce the
single-process results bit for bit: engine level (contiguous aligned blocks + one gather / one 16-byte record per rank) and through
the drop-in `Infer(config, rank=, world=)` (skewed block-cyclic ownership of a growing cache).  SURVEY.md 8e; reference API: infer.py:162-203."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_two(scenario, work, timeout=600, world=2):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="4")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_two_rank_worker.py"), scenario, str(work)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=timeout)[0])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-3000:] for o in outs)
    return json.load(open(os.path.join(work, "result.json")))


def test_two_ranks_engine_sweep_equals_the_single_process_sweep(tmp_path):
    r = _run_two("engine_sweep", tmp_path)
    a = r["aligned"]
    assert a["bounds"][0] == 0 and a["bounds"][1] % 32 == 0
    assert a["overlap_equal"] and a["yaw_equal"], a                  # shards that start at a multiple of 32: same bits
    assert a["decision"] == a["decision_single"] and len(a["decision"]) == 3
    u = r["unaligned"]                                                # 1026 | 1025: rank 1 starts at slot 1026 = 2 mod 32
    assert u["yaw_equal"] and u["max_abs_diff"] <= 2e-6 and u["decision"][0] == u["decision_single"][0]


def test_sharded_infer_equals_the_unsharded_object(tmp_path, fixture_npz):
    from tools import synthetic as S
    frames = 70
    seq = tmp_path / "data" / "07"
    for sub in ("depth", "normal"):
        os.makedirs(seq / sub)
    for i in range(frames):
        s, shift = i % 2, (37 * i) % 900
        np.save(seq / "depth" / ("%06d.npy" % i), np.roll(fixture_npz["range_%d" % s], shift, axis=1))
        np.save(seq / "normal" / ("%06d.npy" % i), np.roll(fixture_npz["normal_%d" % s], shift, axis=1))
    cfg = {"model": dict(S.REFERENCE_MODEL_CFG, inputShape=[64, 900]), "infer_seqs": "07", "data_root_folder": str(tmp_path / "data"),
           "use_depth": True, "use_normals": True, "use_class_probabilities": False, "use_class_probabilities_pca": False,
           "use_intensity": False, "batch_size": 16, "pretrained_weightsfilename": "", "_frames": frames}
    json.dump(cfg, open(tmp_path / "config.json", "w"))
    r = _run_two("infer_api", tmp_path)
    assert r["calls"] == frames and r["mismatch"] == [], r
    # skewed block-cyclic ownership (distributed.frame_owner): even / odd frames alternate inside a block of 32 -> 35 frames each;
    # the local caches' high-water slots: rank 0 holds frame 68 at slot 32 + 4, rank 1 frame 69 at slot 32 + 5
    st = r["stats"]
    assert [s["frames_cached"] for s in st] == [35, 35] and r["local_frames"] == [37, 38], r
    # every scored pair went to the library on the rank's cache rows (feature volume + spectrum + Delta row), none through a
    # replicated or per-pair fallback; the look-ahead computed a Delta row only for the frames the rank owns
    assert all(s["pairs_scored"] > 500 and s["pairs_on_cache_rows"] == s["pairs_scored"] for s in st), st
    assert all(0 < s["ahead_delta_rows"] <= s["frames_cached"] for s in st), st
    f = r["one_rank_failure"]
    assert len(f[0]) == 3 and all("rank(s) [1]" in m for m in f[0]) and all("simulated" in m for m in f[1]), f      # rank 0 / rank 1: both raise, all three calls
    assert r["retry_ok"] is True      # the failed frame did not count as fed: fed again after the repair, then referenced by the next frame
    assert r["order_error"] is True and r["reset_ok"] is True and r["list_reset_ok"] is True
    assert len(r["best"]) == frames // 5 and any(b[0] is not None for b in r["best"])


def test_eight_ranks_replay_the_recorded_demo3_run(tmp_path, fixture_npz):
    """Rehearsal of the multi-GPU form of the reference's real caller: EIGHT ranks (one GPU, gloo / host collectives --
    RCCL refuses two ranks on one device; no N > 1 RCCL run exists) replay the 259 `infer_multiple` calls the reference's own
    demo3_lcd.py made (gated windows of consecutive frame ids, tests/golden/demo_transcript.json) through `Infer(config, rank=, world=8)`,
    some of them as `infer_best_match`: every return value equals the unsharded object's, every scored pair ran on its owner's cache
    rows, and the work shares of THIS run stay within the bound DESIGN.md section 7 quotes for the skewed block-cyclic ownership."""
    from tools import synthetic as S
    frames = 259
    seq = tmp_path / "data" / "07"
    for sub in ("depth", "normal"):
        os.makedirs(seq / sub)
    for i in range(frames):
        s, shift = i % 2, (37 * i) % 900
        np.save(seq / "depth" / ("%06d.npy" % i), np.roll(fixture_npz["range_%d" % s], shift, axis=1))
        np.save(seq / "normal" / ("%06d.npy" % i), np.roll(fixture_npz["normal_%d" % s], shift, axis=1))
    cfg = {"model": dict(S.REFERENCE_MODEL_CFG, inputShape=[64, 900]), "infer_seqs": "07", "data_root_folder": str(tmp_path / "data"),
           "use_depth": True, "use_normals": True, "use_class_probabilities": False, "use_class_probabilities_pca": False,
           "use_intensity": False, "batch_size": 16, "pretrained_weightsfilename": "", "_frames": frames}
    json.dump(cfg, open(tmp_path / "config.json", "w"))
    r = _run_two("infer_demo3", tmp_path, timeout=900, world=8)
    assert r["calls"] == 259 and r["nonempty"] == 83 and r["best_calls"] >= 10 and r["mismatch"] == [], r
    st = r["stats"]
    assert sum(s["frames_cached"] for s in st) == 259 and max(s["frames_cached"] for s in st) - min(s["frames_cached"] for s in st) <= 3, st
    assert all(s["pairs_on_cache_rows"] == s["pairs_scored"] for s in st), st
    scored = np.array([s["pairs_scored"] for s in st], np.float64)
    assert scored.sum() == 1460 and list(scored.astype(int)) == r["owner_work"], (scored, r["owner_work"])
    # work-weighted imbalance of the whole run (max / mean share over the ranks) and the worst single query against its even share
    assert scored.max() / scored.mean() <= 1.31, scored
    assert r["worst_single_query_share"] <= 4.0, r["worst_single_query_share"]      # (a 2-element list on 8 ranks: 1 of 0.25)
