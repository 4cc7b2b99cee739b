"""Worker of tests/test_gpu_two_ranks.py: one of TWO processes that share cuda:0 (gloo rendezvous on 127.0.0.1, host tensors in the
collectives) and run the REAL kernels on their shard.  argv: <scenario> <work dir>.  Rank 0 writes <work dir>/result.json."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from overlapnet_amd import distributed as D  # noqa: E402
from overlapnet_amd.engine import OvnEngine, decode_match  # noqa: E402
from tools import synthetic as S  # noqa: E402


def engine_sweep(work):
    """each rank runs engine.heads on its shard_bounds block of a 2,051-candidate pool and best_match on it; the
    gather must equal the single-process sweep bit for bit, the merged decision its decision."""
    rank, world = dist.get_rank(), dist.get_world_size()
    e = OvnEngine(64, 900, 4, device=0)
    e.load_weights(S.make_test_weights(4, seed=0), S.REFERENCE_MODEL_CFG)
    N = 2051
    g = torch.Generator(device="cuda").manual_seed(77)
    pool = torch.relu(torch.randn((N, 360, 128), device="cuda", generator=g) + 0.1)
    pool *= torch.rand((N, 1, 1), device="cuda", generator=g) * 1.5 + 0.5
    q = torch.relu(torch.randn((1, 360, 128), device="cuda", generator=g) + 0.1)
    qs = e.spectrum(q)
    out = {}
    for align in (D.SLOT_ALIGN, 1):
        lo, hi = D.shard_bounds(N, world, rank, align)
        mine = pool[lo:hi].contiguous()
        r = e.heads(mine, q, spec_l=e.spectrum(mine), spec_r=qs, dcache_l=e.delta_cache(mine))
        rec = e.best_match(r["overlap"], r["yaw"], 0.3, index_offset=lo)
        got = D.gather_scores(r["overlap"], r["yaw"], N, align=align)
        best = D.best_match_sharded(rec)
        if rank == 0:
            full = e.heads(pool, q, spec_l=e.spectrum(pool), spec_r=qs, dcache_l=e.delta_cache(pool))
            f_ov, f_yaw = full["overlap"].cpu(), full["yaw"].cpu()
            key = "aligned" if align > 1 else "unaligned"
            out[key] = {"bounds": [lo, hi], "overlap_equal": bool(torch.equal(got[0], f_ov)), "yaw_equal": bool(torch.equal(got[1], f_yaw)),
                        "max_abs_diff": float((got[0] - f_ov).abs().max()),
                        "decision": list(decode_match(best) or ()), "decision_single": list(decode_match(e.best_match(full["overlap"], full["yaw"], 0.3)) or ())}
    e.close()
    return out


def infer_api(work):
    """`Infer(config, rank=, world=)` against the unsharded object on the same files: every call of a 70-frame
    streaming run (all previous frames, gated subsets, best-match decisions), bit for bit."""
    from overlapnet_amd.infer import Infer
    rank, world = dist.get_rank(), dist.get_world_size()
    cfg = json.load(open(os.path.join(work, "config.json")))
    w = S.make_test_weights(4, seed=0)
    frames = cfg.pop("_frames")
    sh = Infer(json.loads(json.dumps(cfg)), weights=w, rank=rank, world=world)
    ref = Infer(json.loads(json.dumps(cfg)), weights=w) if rank == 0 else None
    rng = np.random.default_rng(5)
    report = {"calls": 0, "mismatch": [], "local_frames": None, "best": []}
    for i in range(frames):
        if i % 7 == 3 and i > 4:
            refs = sorted(rng.choice(i, size=min(i, 9), replace=False).tolist())      # a gated subset
        elif i % 11 == 5:
            refs = []
        else:
            refs = list(range(i))
        if i % 5 == 4:
            a = sh.infer_best_match(i, refs, 0.3)
            if rank == 0:
                b = ref.infer_best_match(i, refs, 0.3)
                report["best"].append([list(a) if a else None, list(b) if b else None])
                if a != b:
                    report["mismatch"].append(["best", i])
        else:
            a = sh.infer_multiple(i, refs)
            if rank == 0:
                b = ref.infer_multiple(i, refs)
                same = (a is None and b is None) or (a is not None and b is not None and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
                                                     and a[0].shape == b[0].shape and a[1].dtype == b[1].dtype)
                if not same:
                    report["mismatch"].append(["multiple", i])
        report["calls"] += 1
    counts = [None, None]
    dist.all_gather_object(counts, [len(sh.feature_volumes), dict(sh.sharded_stats)])
    report["local_frames"] = [c[0] for c in counts]
    report["stats"] = [c[1] for c in counts]
    try:
        sh.infer_multiple(frames + 3, [0])           # out of order: refused in sharded mode
        report["order_error"] = False
    except Exception as ex:
        report["order_error"] = "order" in str(ex)
    # the reference's way to reset the cache works in sharded mode too: frames start again at 0
    sh.feature_volumes = []
    a0 = sh.infer_multiple(0, [])
    a1 = sh.infer_multiple(1, [0])
    if rank == 0:
        ref.feature_volumes = []
        ref.infer_multiple(0, [])
        b1 = ref.infer_multiple(1, [0])
        report["reset_ok"] = bool(a0 is None and np.array_equal(a1[0], b1[0]) and np.array_equal(a1[1], b1[1]) and a1[0].shape == ())
    # a failure of ONE rank's local work (here: the leg of rank 1, the OWNER of frame 3, raises) reaches every rank through the
    # collective's payload: both raise, nobody is left waiting in the all-gather -- also when the reference list is empty
    # (no scores to exchange: a 16-byte status record instead) -- and the frame does NOT count as fed on any rank: its
    # owner's slot was never written, so the same frame is fed again once the cause is repaired, and later frames may refer to it
    a2 = sh.infer_multiple(2, [0, 1])
    good_leg = sh._leg_device
    if rank == 1:
        def boom(names):
            raise Exception("Could not read depth image (simulated, rank 1 only)")
        sh._leg_device = boom
    sh._stream_ahead = False
    sh._drop_ahead()
    raised = [None, None, None]
    for k, call in enumerate((lambda: sh.infer_multiple(3, [0, 1, 2]), lambda: sh.infer_best_match(3, [0, 1, 2], 0.3),
                              lambda: sh.infer_multiple(3, []))):
        try:
            call()
            raised[k] = "returned"
        except Exception as ex:
            raised[k] = str(ex)[:80]
    both = [None, None]
    dist.all_gather_object(both, raised)
    report["one_rank_failure"] = both
    sh._leg_device = good_leg
    a3 = sh.infer_multiple(3, [0, 1, 2])            # the same frame again
    a4 = sh.infer_multiple(4, [0, 1, 2, 3])         # ... and a frame that refers to it
    if rank == 0:
        b2, b3, b4 = ref.infer_multiple(2, [0, 1]), ref.infer_multiple(3, [0, 1, 2]), ref.infer_multiple(4, [0, 1, 2, 3])
        report["retry_ok"] = bool(all(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) for a, b in ((a2, b2), (a3, b3), (a4, b4))))
    # the cache rebuilt from a LIST of volumes in sharded mode (the reference's `infer.feature_volumes = [...]`): every rank keeps the
    # frames it owns at their slots (one batched upload), and the next sweep over all of them equals the unsharded object's
    vols = [None]
    if rank == 0:
        vols = [np.asarray(ref.feature_volumes)]          # frames 0 .. 4 of the unsharded cache: (5, 1, 360, 128)
    dist.broadcast_object_list(vols, src=0)
    sh.feature_volumes = list(vols[0])
    a5 = sh.infer_multiple(5, [0, 1, 2, 3, 4])
    if rank == 0:
        b5 = ref.infer_multiple(5, [0, 1, 2, 3, 4])
        report["list_reset_ok"] = bool(np.array_equal(a5[0], b5[0]) and np.array_equal(a5[1], b5[1]))
    sh.close()
    if ref is not None:
        ref.close()
    return report


def infer_demo3(work):
    """the reference's REAL caller at world 8 -- the calls its own demo3_lcd.py made on a 259-frame run (recorded in
    tests/golden/demo_transcript.json: `infer_multiple(frame, gated reference list)`, windows of consecutive frame ids, 83 non-empty
    lists) replayed through `Infer(config, rank=, world=)` on every rank; every 7th non-empty query goes through `infer_best_match`
    instead.  Rank 0 replays the same calls on the unsharded object and compares every return value; every rank reports the pairs it
    scored (`sharded_stats`) so that the test can assert the work shares of the run itself."""
    from overlapnet_amd.infer import Infer
    from overlapnet_amd import distributed as D
    rank, world = dist.get_rank(), dist.get_world_size()
    cfg = json.load(open(os.path.join(work, "config.json")))
    w = S.make_test_weights(4, seed=0)
    cfg.pop("_frames")
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "demo_transcript.json")) as fh:
        events = [e for e in json.load(fh)["demo3"] if e.get("event") == "infer_multiple"]
    sh = Infer(json.loads(json.dumps(cfg)), weights=w, rank=rank, world=world)
    ref = Infer(json.loads(json.dumps(cfg)), weights=w) if rank == 0 else None
    report = {"calls": 0, "mismatch": [], "nonempty": 0, "best_calls": 0, "owner_work": None}
    work_by_rank = np.zeros(world, np.int64)
    worst_share = 0.0
    for e in events:
        cur, refs = int(e["cur"]), [int(x) for x in e["refs"]]
        best = bool(refs) and report["nonempty"] % 7 == 3
        if refs:
            report["nonempty"] += 1
            cnt = np.bincount(D.frame_owner(np.asarray(refs), world), minlength=world)
            work_by_rank += cnt
            worst_share = max(worst_share, float(cnt.max()) / (len(refs) / world))
        if best:
            a = sh.infer_best_match(cur, refs, 0.3)
            report["best_calls"] += 1
            if rank == 0:
                b = ref.infer_best_match(cur, refs, 0.3)
                if a != b:
                    report["mismatch"].append(["best", cur])
        else:
            a = sh.infer_multiple(cur, refs)
            if rank == 0:
                b = ref.infer_multiple(cur, refs)
                same = (a is None and b is None) or (a is not None and b is not None and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
                                                     and a[0].shape == b[0].shape and a[1].dtype == b[1].dtype)
                if not same:
                    report["mismatch"].append(["multiple", cur])
        report["calls"] += 1
    stats = [None] * world
    dist.all_gather_object(stats, [len(sh.feature_volumes), dict(sh.sharded_stats)])
    report["stats"] = [s_[1] for s_ in stats]
    report["local_frames"] = [s_[0] for s_ in stats]
    report["owner_work"] = work_by_rank.tolist()                     # pairs each rank owns over the whole run (host arithmetic)
    report["worst_single_query_share"] = worst_share                 # max over queries of (largest rank share / even share)
    sh.close()
    if ref is not None:
        ref.close()
    return report


def main():
    scenario, work = sys.argv[1], sys.argv[2]
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
    try:
        out = {"engine_sweep": engine_sweep, "infer_api": infer_api, "infer_demo3": infer_demo3}[scenario](work)
        if dist.get_rank() == 0:
            json.dump(out, open(os.path.join(work, "result.json"), "w"))
        dist.barrier()
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
