#!/usr/bin/env python3
"""Benchmark of the OverlapNet hot path on MI355X -- BASELINE.json metric: scan-pairs/s on 64x900 range images.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--pool P] [--channels C]     (N > 1: spawns one rank per GPU itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One STEP = one loop-closure query the way the reference's `Infer.infer_multiple` runs it
(src/two_heads/infer.py:162-203): the leg over the query scan (1 scan, 64x900xC, already in HBM) plus BOTH
heads of that query against the candidate feature volumes cached in HBM ("warm" sweep: candidates' legs ran
when they were the current frame, infer.py:184-185).

N = 1 (default): P = 1024 candidates, C = 4 = BASELINE.json configs[1] ("batched 1-vs-1024 pairs, 64x900 depth+normals").
`value` is that warm sweep for a STREAM of queries: every step enqueues one query leg and one 1024-pair head sweep, the leg (and
spectrum) of query k + 1 on a second context / stream beside the head kernels of query k (overlapnet_amd.engine.QueryAhead;
`--serial-query` puts the leg in front of its own heads on one stream instead).  The same run then measures, each in its own
timed region and reported as sub-records of the ONE JSON line: `warm_serial` (that one-stream order), `fp32_mode` (every contraction on the fp32 matrix cores), `cold` (candidate legs inside the step),
`fullstack` (raw clouds -> projection -> legs -> heads), `corr_head` (the HBM-bound correlation head alone at N = 1024 and
16384), `infer_api` (BASELINE configs[2]: the 1101-frame loop-closure sweep through `Infer.infer_multiple`), `latency` (one query
against N = 1 [configs[0]], 16, 100, 256 candidates), and the accuracy of the timed configuration over ALL pairs against the
committed fp64-oracle outputs.

N > 1: BASELINE.json configs[3], STRONG scaling: one 1-vs-100000 synthetic candidate pool (`--pool-total`) sharded in
contiguous blocks over the ranks (overlapnet_amd.distributed.shard_bounds), feature volumes generated on the device
(SURVEY.md 8d), one RCCL gather of (overlap, yaw) to rank 0 per step; value = pool_total * steps / elapsed.
`--pool-total 0` gives the weak-scaling variant (P candidates per rank).  Prints ONE JSON line on rank 0.

`--rehearsal` (or OVN_BENCH_REHEARSAL=1) with N > 1: the SAME command path -- launcher, one rank per "GPU", shard bounds, per-step
gather to rank 0, every rank's self-check, rank-0-only JSON -- with all N ranks on ONE device and gloo / host tensors in the
collectives (RCCL refuses two ranks on one device); rank 0 then evaluates the whole pool in one process and requires the gathered
(overlap, yaw) to be `torch.equal` to it.  The line is marked `"rehearsal": true` and carries no pairs/s (`value` is null): N
processes time-sharing one GPU say nothing about scaling.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from overlapnet_amd import distributed as D  # noqa: E402
from tools import synthetic as S  # noqa: E402
from overlapnet_amd.engine import OvnEngine  # noqa: E402

# algorithmic work of the dominant kernel, SURVEY.md section 8a row a7: c_conv1 8640 x 1920 x 64 and c_conv2 576 x 960 x 128
# multiply-adds per pair, FLOP = 2 * MAC.  fp32 mode: one fused kernel does both; f16x3 mode: the timed kernel
# (delta_c1_f16x3_kernel) is DeltaLayer + c_conv1 only, c_conv2 is its own kernel (`kernels.delta_c2`, `delta_total_ms`).
DELTA_C1_FLOP_PER_PAIR = 2 * 8640 * 1920 * 64
DELTA_C2_FLOP_PER_PAIR = 2 * 576 * 960 * 128
CAND_BYTES_PER_PAIR = 184_320 + 8           # candidate feature volume read once + (overlap, yaw) written
SPEC_BYTES_PER_PAIR = 188_416 + 4           # what the spectral form streams: one cached spectrum (128 x 368 f32) + yaw
LEG_FLOP_PER_SCAN = {1: 1637.5e6, 4: 1733.2e6, 5: 1765.1e6}
PEAK_F32_MFMA_TFLOPS = 157.3                # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 dense peak
PEAK_16BIT_MFMA_TFLOPS = 2500.0             # MI355X_MICROARCH.md: dense bf16/fp16 MFMA peak (not the 2:1-sparse figure)
SUSTAINED_16BIT_MFMA_TFLOPS = 1850.0        # what a BARE fp16 MFMA loop on random register operands sustains on this chip (power management
                                            # holds it at 2.09 GHz; 2272 TF on all-zero operands): profiles/r5_mfma_power.txt, tools/experiments/mfma_power.hip
PEAK_HBM_BPS = 8.0e12

HEAD_KERNEL = {
    "f32": ("delta_c12_kernel (DeltaLayer+c_conv1+c_conv2, fp32 MFMA)", PEAK_F32_MFMA_TFLOPS, "delta_c12_kernel",
            DELTA_C1_FLOP_PER_PAIR + DELTA_C2_FLOP_PER_PAIR, "fp32 matrix cores, one MFMA per product"),
    "f16x3": ("delta_c1_f16x3_kernel (DeltaLayer in min form + c_conv1, fp16 MFMA)", PEAK_16BIT_MFMA_TFLOPS, "delta_c1_f16x3",
              DELTA_C1_FLOP_PER_PAIR,
              "achieved counts ALGORITHMIC flops of DeltaLayer + c_conv1 (93.8 % of the Delta head's flops; c_conv2 and the linear "
              "terms run in delta_c2 / delta_prep, see kernels); the 3-term split issues 3 MFMA flops per algorithmic flop WALKED, and "
              "a 1-vs-N sweep walks only the channels that are alive in the query (k_walk_frac of the 128, exact: query_live_channels; "
              "dense_walk_pairs_per_s = the same step with ovn_set_head_compaction 0)"),
}
DTYPE_LABEL = {
    "f32": "f32",
    "f16x3": "f32 (scaled 3-term fp16 split on the fp16 MFMA: 22 significand bits per operand, fp32 accumulate)",
}


def _committed_traffic(kernel_prefix: str):
    """(HBM bytes per launch, profile tag) of the kernel whose name starts with `kernel_prefix` from the NEWEST committed
    rocprofv3 PMC summary under profiles/ (tools/summarize_rocprof.py) -- (None, None) if no summary names it."""
    import glob
    import re

    def tag_key(path):   # r1 < r1b < ... < r1j < r2a < r2f: (round, letter suffix); file times do not survive a checkout
        m = re.match(r"r(\d+)([a-z]*)_", os.path.basename(path))
        return (int(m.group(1)), m.group(2)) if m else (-1, "")
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_rocprof_summary.json")), key=tag_key)
    for f in reversed(files):
        try:
            t = json.load(open(f)).get("hbm_traffic_per_launch", {})
        except Exception:
            continue
        hits = [v["total_bytes"] for k, v in t.items() if k.startswith(kernel_prefix)]
        if hits:
            return max(hits), "profiles/" + os.path.basename(f)
    return None, None


def _measured_traffic(kernel_prefix: str, extra_args, timeout_s: float = 150.0):
    """HBM bytes per launch of the dominant kernel measured NOW: two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE do not fit one
    pass on gfx950; kernel trace only, no other trace domain) over a short run of this same program, read back from the rocpd
    database; FETCH_SIZE doubled (MI355X_MICROARCH.md, HBM section: gfx950 tallies the 128-B requests of wide coalesced reads at
    64 B), both counters in KiB.  Returns (bytes, {"read": .., "write": ..}) or (None, reason)."""
    import glob
    import shutil
    import sqlite3
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.isfile("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, "rocprofv3 not found"
    got = {}
    tmp = tempfile.mkdtemp(prefix="ovn_traffic_", dir="/tmp")
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            out_dir = os.path.join(tmp, ctr)
            cmd = [exe, "--kernel-trace", "--pmc", ctr, "-d", out_dir, "-o", "bench", "--", sys.executable, os.path.abspath(__file__),
                   "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-extras", "--no-walk-stats", "--traffic", "none"] + list(extra_args)
            env = dict(os.environ, TMPDIR="/tmp")
            try:
                p = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
                try:
                    p.wait(timeout=timeout_s)
                except subprocess.TimeoutExpired:
                    import signal
                    os.killpg(p.pid, signal.SIGKILL)       # the process group WE started, nothing else
                    p.wait()
                    return None, "rocprofv3 pass %s timed out after %d s" % (ctr, timeout_s)
            except OSError as e:
                return None, "rocprofv3: %s" % e
            dbs = glob.glob(os.path.join(out_dir, "**", "*.db"), recursive=True)
            if not dbs:
                return None, "rocprofv3 pass %s left no database (rc %s)" % (ctr, p.returncode)
            vals = []
            for db in dbs:
                try:
                    con = sqlite3.connect(db)
                    q = "select value from counters_collection where counter_name = ? and kernel_name like ? order by dispatch_id"
                    vals += [r[0] for r in con.execute(q, (ctr, "%" + kernel_prefix + "%"))]
                    con.close()
                except sqlite3.Error as e:
                    return None, "rocpd database: %s" % e
            if not vals:
                return None, "no %s rows for %s" % (ctr, kernel_prefix)
            tail = vals[-3:]                               # the timed steps' launches
            got[ctr] = sum(tail) / len(tail) * 1024.0      # KiB -> B
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    rd, wr = 2.0 * got["FETCH_SIZE"], got["WRITE_SIZE"]
    return rd + wr, {"read_bytes": rd, "write_bytes": wr}


def rocprof_traffic(kernel_prefix: str, mode: str, extra_args):
    """-> (bytes per launch or None, where the figure comes from).  mode: 'measure' (PMC passes now, committed profile if they
    fail), 'committed', 'none'."""
    if mode == "none":
        return None, "not collected (--traffic none)"
    if mode == "measure":
        b, info = _measured_traffic(kernel_prefix, extra_args)
        if b is not None:
            return b, dict(info, source="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes inside this run (2 x FETCH_SIZE, gfx950 correction)")
        note = info
    else:
        note = "not measured (--traffic committed)"
    b, src = _committed_traffic(kernel_prefix)
    return b, {"source": src, "note": note}


def cpu_baseline_preprocess(n_scans: int = 3):
    """the reference's NumPy preprocessing (`range_projection` + `gen_normal_map`, utils.py:59-186)
    timed beside the HIP scatter, on this host, single thread as shipped.  kind 'port': the oracle's restatement in the reference's
    STRUCTURE -- vectorised NumPy projection (utils.py:75-132 is vectorised too) and the per-pixel Python double loop of
    `gen_normal_map` (utils.py:149-173; `oracle.gen_normal_map_literal`, bit-identical to the shipped normal images); the oracle's
    vectorised normal map is reported next to it (`vectorised_scans_per_s`: what a NumPy user could have had).  kind 'reference':
    the reference's own `utils.py` when OVERLAPNET_REFERENCE names a checkout of PRBonn/OverlapNet (never set on the bench box)."""
    from oracle import overlapnet_oracle as O
    fx = S.load_fixture_images()
    clouds = [S.fullstack_cloud(fx, i) for i in range(n_scans)]
    kind, proj, normals = "port", O.range_projection, O.gen_normal_map_literal
    ref_root = os.environ.get("OVERLAPNET_REFERENCE")
    if ref_root and os.path.isfile(os.path.join(ref_root, "src", "utils", "utils.py")):
        try:
            import importlib.util
            spec = importlib.util.spec_from_file_location("_ref_utils", os.path.join(ref_root, "src", "utils", "utils.py"))
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            kind, proj, normals = "reference", mod.range_projection, mod.gen_normal_map
        except Exception:
            pass
    proj(clouds[0])
    t0 = time.perf_counter()
    for c in clouds:
        r = proj(c)
    t_proj = (time.perf_counter() - t0) / len(clouds)
    t0 = time.perf_counter()
    normals(r[0], r[1])
    t_norm = time.perf_counter() - t0
    t0 = time.perf_counter()
    O.gen_normal_map(r[0], r[1])
    t_vec = time.perf_counter() - t0
    return {"value": 1.0 / (t_proj + t_norm), "unit": "scans/s", "cores": 1, "kind": kind,
            "vectorised_scans_per_s": 1.0 / (t_proj + t_vec),
            "sample": "%d KITTI-sized clouds through range_projection (%.4f s/scan) + ONE gen_normal_map as shipped, a per-pixel Python loop "
                      "(%.3f s/scan; vectorised restatement %.4f s)" % (len(clouds), t_proj, t_norm, t_vec)}


def cpu_baseline(channels: int, pool: int):
    """The CPU restatement oracle (PyTorch-CPU fp32, structured like the reference: leg model, then head
    model in batches of 16 with the 360x360x128 Delta tensor materialised) timed on this host's cores on
    a bounded sample: 3 legs + 32 pairs, extrapolated to the 1-leg + `pool`-pairs step."""
    from oracle import overlapnet_oracle as O
    ncpu = os.cpu_count() or 1
    w = S.make_test_weights(channels, seed=0)
    imgs = S.candidate_images(3, channels, seed=5)
    fv0 = O.leg_forward(imgs[:1], w, S.REFERENCE_MODEL_CFG, np.float32)
    # use the thread count that serves this workload best on this host (all cores is not it on a 256-thread box)
    best = None
    for th in sorted({ncpu, min(ncpu, 64), min(ncpu, 16)}, reverse=True):
        torch.set_num_threads(th)
        O.heads_forward(fv0, fv0, w, dtype=np.float32)
        t0 = time.perf_counter()
        O.heads_forward(fv0, fv0, w, dtype=np.float32)
        dt = time.perf_counter() - t0
        if best is None or dt < best[1]:
            best = (th, dt)
    cores = best[0]
    torch.set_num_threads(cores)
    t0 = time.perf_counter()
    fv = O.leg_forward(imgs, w, S.REFERENCE_MODEL_CFG, np.float32)
    t_leg = (time.perf_counter() - t0) / imgs.shape[0]
    n_pairs = 32
    li = np.arange(n_pairs) % 3
    ri = np.zeros(n_pairs, int)
    t0 = time.perf_counter()
    for b in range(0, n_pairs, 16):
        O.heads_forward(fv[li[b:b + 16]], fv[ri[b:b + 16]], w, dtype=np.float32)
    t_pair = (time.perf_counter() - t0) / n_pairs
    step_s = t_leg + pool * t_pair
    return {"value": pool / step_s, "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": "oracle fp32 (PyTorch-CPU): 3 legs + 32 head pairs in batches of 16 timed, extrapolated to "
                      "1 leg + %d pairs; leg %.3f s/scan, heads %.4f s/pair" % (pool, t_leg, t_pair)}


def relaunch_command(argv, gpus, port=None, python=None):
    """The command line `python bench.py --gpus N ...` re-executes itself with when it was started without a launcher:
    the one the driver uses for N > 1 (one rank per GPU of ONE node, rendezvous on 127.0.0.1)."""
    if port is None:
        port = int(os.environ.get("MASTER_PORT", "0")) or (29500 + os.getpid() % 2000)
    return [python or sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(int(gpus)),
            "--master-addr", "127.0.0.1", "--master-port", str(int(port)), os.path.abspath(__file__)] + list(argv)


def relaunch_env(env):
    e = dict(env)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the host driver only supports dmabuf IPC (RCCL across processes)
    e.setdefault("OMP_NUM_THREADS", "8")
    return e


def timed(step, warmup, steps, eng, use_dist, dev, side_eng=None):
    """W untimed warm-up steps, then exactly K steps bracketed by barrier + synchronize on both sides; max over ranks.
    Returns (elapsed_s, per-kernel HIP-event profile, last step result).  `side_eng`: a second library context whose kernels
    (the query leg of QueryAhead) belong to the same steps; its event times are merged into the profile."""
    res = None
    for _ in range(warmup):
        res = step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    eng.profile_begin()
    if side_eng is not None:
        side_eng.profile_begin()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        res = step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    prof = eng.profile_end()
    if side_eng is not None:
        for k, (ms, cnt) in side_eng.profile_end().items():
            prof[k] = (prof[k][0] + ms, prof[k][1] + cnt)
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if dist.get_backend() != "gloo" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed, prof, res


def kernel_table(prof):
    return {k: {"ms_per_launch": v[0] / v[1], "launches": v[1]} for k, v in prof.items() if v[1]}


def golden_accuracy(ov, yw, wset="glorot"):
    """All pairs of the sweep against the committed fp64-oracle outputs (tests/golden/make_parity_sweep_golden.py)."""
    path = os.path.join(ROOT, "tests", "golden", "parity_sweep_%s.npz" % wset)
    with np.load(path) as z:
        g = {k: z[k] for k in z.files}
    d = np.abs(ov.astype(np.float64) - g["overlap"])
    bad = yw != g["yaw"]
    return {"overlap_mae_vs_oracle": float(d.mean()), "overlap_maxerr_vs_oracle": float(d.max()),
            "overlap_p99err_vs_oracle": float(np.percentile(d, 99)),
            "yaw_exact_rate": float(np.mean(~bad)),
            "yaw_mismatch_max_oracle_top2_gap": float(g["corr_top2_gap"][bad].max()) if bad.any() else 0.0,
            "accuracy_pairs": int(ov.size),
            "accuracy_scope": "images -> leg -> heads on the GPU vs the fp64 oracle on the same images, every pair of the sweep "
                              "(oracle outputs: tests/golden/parity_sweep_%s.npz)" % wset}


def trained_like_record(C, P, dev, pool_imgs, query_ring, query_img, steps, flop_per_pair, peak):
    """The timed warm step (streamed queries, 1-vs-P, cached candidates) under the trained-like weight set, with and without the
    dead-channel compaction; accuracy of query 0 over all pairs against the committed fp64 oracle outputs of that set."""
    from overlapnet_amd.engine import QueryAhead
    wt = S.make_trained_like_weights(C)
    e2 = OvnEngine(64, 900, C, device=dev.index)
    e2.load_weights(wt, S.REFERENCE_MODEL_CFG)
    qa2 = None
    try:
        cands = torch.empty((P, 360, 128), dtype=torch.float32, device=dev)
        for s0 in range(0, P, 128):
            e2.leg(pool_imgs[s0:s0 + 128], out=cands[s0:s0 + 128])
        spec, dc = e2.spectrum(cands), e2.delta_cache(cands)
        qa2 = QueryAhead(e2, wt, S.REFERENCE_MODEL_CFG)
        pos = [0]

        def nxt():
            pos[0] = (pos[0] + 1) % len(query_ring)
            return query_ring[pos[0]]
        qa2.submit(nxt())

        def step():
            qa2.submit(nxt())
            fv, sp = qa2.take()
            r = e2.heads(cands, fv, spec_l=spec, spec_r=sp, dcache_l=dc)
            return r["overlap"], r["yaw"]
        rec = {"weights": "tools/synthetic.make_trained_like_weights (leg gain 1.6, O(1) biases, Dense gain 10: logits -20 .. 18)"}
        for name, on in (("compacted", True), ("dense_walk", False)):
            e2.set_head_compaction(on)
            el, pr, _ = timed(step, 2, steps, e2, False, dev, side_eng=qa2.side)
            c12 = pr["delta_c12"][0] / max(pr["delta_c12"][1], 1)
            rec[name] = {"value": P * steps / el, "unit": "pairs/s", "ms_per_step": 1e3 * el / steps, "delta_c12_ms": c12,
                         "frac": flop_per_pair * P / (c12 * 1e-3) / 1e12 / peak}
        e2.set_head_compaction(True)
        walks, lives = [], []
        for qi in query_ring:
            fq = e2.leg(qi)
            e2.heads(cands[:32], fq, spec_l=spec[:32], spec_r=e2.spectrum(fq), dcache_l=dc[:32])
            st = e2.head_walk_stats()
            walks.append(st["k_walk_frac"])
            lives.append(st["live_channels"])
        rec["k_walk_frac"] = sum(walks) / len(walks)
        rec["k_walk_by_query"] = {"live_channels": lives, "k_walk_frac": walks}
        qa2.take()
        fq = e2.leg(query_img)
        r = e2.heads(cands, fq, spec_l=spec, spec_r=e2.spectrum(fq), dcache_l=dc)
        torch.cuda.synchronize()
        if P == 1024 and C == 4 and os.path.isfile(os.path.join(ROOT, "tests", "golden", "parity_sweep_trained_like.npz")):
            ga = golden_accuracy(r["overlap"].float().cpu().numpy(), r["yaw"].cpu().numpy(), "trained_like")
            rec.update({k: ga[k] for k in ("overlap_mae_vs_oracle", "overlap_maxerr_vs_oracle", "yaw_exact_rate", "accuracy_pairs")})
        return rec
    finally:
        if qa2 is not None:
            qa2.close()
        e2.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pool", type=int, default=1024, help="candidate feature volumes resident per GPU (N = 1, or weak scaling)")
    ap.add_argument("--pool-total", type=int, default=None,
                    help="N > 1: size of the ONE candidate pool sharded over the ranks (default 100000 = BASELINE configs[3]); "
                         "0 = weak scaling with --pool candidates per rank")
    ap.add_argument("--channels", type=int, default=4, help="4 = depth+normals (network.yml), 1 = depth, 5 = +intensity")
    ap.add_argument("--head-precision", default="f16x3", choices=["f32", "f16x3"],
                    help="Delta-head contraction arithmetic (fp32 storage/accumulation in both modes): scaled 3-term fp16 split "
                         "(default) or fp32 MFMA")
    ap.add_argument("--leg-precision", default=None, choices=["f32", "f16x3"], help="leg convolution arithmetic (default: the engine's)")
    ap.add_argument("--mode", default="warm", choices=["warm", "cold", "fullstack"],
                    help="what `value` times -- warm (default, the BASELINE metric): candidate features cached, step = query leg "
                         "+ heads; cold: step also runs the legs of all candidates from images resident in HBM; fullstack: step "
                         "starts from raw point clouds (projection + normals on the GPU)")
    ap.add_argument("--corr", default="spectral", choices=["spectral", "direct"],
                    help="correlation head: spectral form on cached candidate spectra (default) or direct Gram form")
    ap.add_argument("--force-dist", action="store_true", help="initialise RCCL and run the gather even with one rank (path check)")
    ap.add_argument("--no-delta-cache", action="store_true", help="warm sweep without the candidates' Delta cache rows (round-2 behaviour)")
    ap.add_argument("--no-compaction", action="store_true",
                    help="walk all 128 feature channels in the Delta contraction even where the query's are dead (ovn_set_head_compaction 0)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--serial-query", action="store_true", help="warm mode: the query leg on the heads' stream, in front of them (no QueryAhead)")
    ap.add_argument("--no-extras", action="store_true", help="skip the fp32_mode / cold / fullstack / corr_head sub-records")
    ap.add_argument("--no-walk-stats", action="store_true",
                    help="skip the untimed 32-pair sweeps that read the K walk of each query of the stream back (profiling runs: every "
                         "launch of the trace is then a full sweep)")
    ap.add_argument("--rehearsal", action="store_true", default=os.environ.get("OVN_BENCH_REHEARSAL", "") == "1",
                    help="N > 1 ranks on ONE device, gloo / host collectives: rehearses the driver's multi-GPU command path (no pairs/s claim)")
    ap.add_argument("--traffic", default="measure", choices=["measure", "committed", "none"],
                    help="roofline.traffic: two rocprofv3 PMC passes over a short run of this program (default, N = 1 warm mode; falls back "
                         "to the newest committed profile), the committed profile, or nothing")
    ap.add_argument("--accuracy-pairs", type=int, default=12, help="pairs checked against a LIVE fp64 oracle when no committed "
                                                                  "oracle outputs exist for the configuration (untimed)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: become the launcher -- one rank per GPU under torch.distributed.run, rank 0 prints the
        # ONE JSON line on the inherited stdout (one process per GPU, RCCL over xGMI)
        cmd = relaunch_command(sys.argv[1:], args.gpus)
        raise SystemExit(subprocess.call(cmd, env=relaunch_env(os.environ)))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    visible = torch.cuda.device_count()
    rehearsal = bool(args.rehearsal and world > 1)
    if rehearsal:
        if visible < 1:
            raise SystemExit("bench.py --rehearsal: needs one GPU, none visible (rank %d)" % rank)
        local_rank = 0                      # every rank on the ONE device
    elif visible < max(world, 1) or local_rank >= visible:
        raise SystemExit("bench.py: %d GPUs needed, %d visible (rank %d)" % (world, visible, rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # RCCL prints its version banner (and gloo its "[Gloo] Rank r is connected ..." lines) on STDOUT (C stdio, fully buffered when
        # stdout is a pipe) at communicator creation; this program's stdout is ONE JSON line, so fd 1 points at stderr until the
        # first collective has run and C stdio is flushed
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            if rehearsal:     # all ranks on ONE device: host tensors in the collectives (distributed._comm_device)
                dist.init_process_group("gloo", rank=rank, world_size=world)
            else:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            try:
                import ctypes
                ctypes.CDLL(None).fflush(None)
            except Exception:
                pass
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)

    C = args.channels
    pool_total = args.pool_total
    if pool_total is None:
        pool_total = 100000 if world > 1 else 0
    strong = pool_total > 0
    if strong and args.mode != "warm":
        raise SystemExit("--pool-total (one pool sharded over the ranks) is a warm sweep; use --pool-total 0 with --mode %s" % args.mode)
    if strong:
        # blocks that start at a multiple of 32: a candidate keeps its pool slot modulo 32 -> the bits of the unsharded sweep
        lo, hi = D.shard_bounds(pool_total, world, rank, D.SLOT_ALIGN)
        P = hi - lo
        n_total = pool_total
    else:
        P = args.pool
        n_total = P * world

    eng = OvnEngine(64, 900, C, device=local_rank)
    w = S.make_test_weights(C, seed=0)
    eng.load_weights(w, S.REFERENCE_MODEL_CFG)
    eng.set_head_precision(args.head_precision)
    eng.set_head_compaction(not args.no_compaction)
    if args.leg_precision:
        eng.set_leg_precision(args.leg_precision)
    leg_precision = eng.leg_precision

    # ---- untimed setup: candidate pool -> feature volumes resident in HBM ----
    fx = S.load_fixture_images()
    flags = S.flags_of(C)
    cands = torch.empty((P, 360, 128), dtype=torch.float32, device=dev)
    pool_imgs = all_imgs = None
    if strong:
        # feature volumes generated on the device, relu(N(0.1, 1)) (~46 % zeros like leg outputs),
        # Philox seed 1234 + rank; never 92 GB of images
        g = torch.Generator(device=dev).manual_seed(1234 + rank)
        for s in range(0, P, 4096):
            n = min(4096, P - s)
            cands[s:s + n] = torch.relu(torch.randn((n, 360, 128), device=dev, generator=g) + 0.1)
    else:
        keep_imgs = (args.mode == "cold") or (world == 1 and not args.no_extras)
        if keep_imgs:   # P candidate images + one slot for the query image at the end (the cold step's leg input, resident in HBM)
            all_imgs = torch.empty((P + 1, 64, 900, C), dtype=torch.float32, device=dev)
            pool_imgs = all_imgs[:P]
        for s, imgs in S.sweep_pool_images(P, C, rank, fx):
            timg = torch.from_numpy(imgs).to(dev)
            if pool_imgs is not None:
                pool_imgs[s:s + timg.shape[0]].copy_(timg)
            eng.leg(timg, out=cands[s:s + timg.shape[0]])
    query_img = torch.from_numpy(S.sweep_query_image(C, fx)).to(dev)
    # the stream of queries of the warm step: EIGHT DIFFERENT SCANS taken in turn -- the query scan and seven scans drawn like the pool's
    # candidates (both shipped scans, other column shifts, their own depth noise; another seed than the pool's), each through the leg
    # inside its step.  Their live-channel counts / K walks are reported per query (roofline.k_walk_by_query); the accuracy block
    # evaluates query 0 in a step of its own
    ring_np = S.candidate_images(8, C, seed=4321, fixture=fx)[1:]
    query_ring = [query_img] + [torch.from_numpy(np.ascontiguousarray(np.roll(ring_np[k], 113 * (k + 1), axis=1))[None]).to(dev)
                                for k in range(7)]
    ring_pos = [0]

    def next_query():
        ring_pos[0] = (ring_pos[0] + 1) % len(query_ring)
        return query_ring[ring_pos[0]]
    if pool_imgs is not None:
        all_imgs[P:].copy_(query_img)
    query_fv = torch.empty((1, 360, 128), dtype=torch.float32, device=dev)

    def make_raw():
        # raw scans resident in HBM: the two shipped KITTI scans rotated about z, candidate i by (37 i mod 900) columns; built on the
        # host by the recipe the committed oracle outputs were made from (tools/synthetic.fullstack_cloud), untimed
        pts, offs = [], [0]
        for i in range(P + 1):
            q = torch.from_numpy(S.fullstack_cloud(fx, i)).to(dev)
            pts.append(q)
            offs.append(offs[-1] + q.shape[0])
        return (torch.cat(pts).contiguous(), torch.tensor(offs, dtype=torch.int64, device=dev), max(p.shape[0] for p in pts))

    spectral = args.corr == "spectral"
    cand_spec = eng.spectrum(cands) if spectral else None          # cached per candidate, like its feature volume
    cand_dc = eng.delta_cache(cands) if (spectral and args.head_precision == "f16x3" and not args.no_delta_cache) else None
    query_spec = torch.empty((1, 128, eng.SPEC_W), dtype=torch.float32, device=dev)
    all_fv = None
    torch.cuda.synchronize()

    def finish(r):
        if use_dist:
            return D.gather_scores(r["overlap"], r["yaw"], n_total, align=D.SLOT_ALIGN if strong else 1)      # rank 0: the pool in order
        return r["overlap"], r["yaw"]

    def step_warm_serial(img=None):
        eng.leg(next_query() if img is None else img, out=query_fv)
        if spectral:
            eng.spectrum(query_fv, out=query_spec)
            return finish(eng.heads(cands, query_fv, spec_l=cand_spec, spec_r=query_spec, dcache_l=cand_dc))
        return finish(eng.heads(cands, query_fv))

    # Streaming queries (the demo3 loop over a recorded sequence): the leg + spectrum of query k + 1 run on a second context and
    # stream beside the head kernels of query k (overlapnet_amd.engine.QueryAhead).  Every step still enqueues ONE query leg and
    # ONE 1024-pair head sweep; the pipeline is primed before the warm-up steps, so the K timed steps hold K legs and K sweeps.
    qa = None
    if args.mode == "warm" and not args.serial_query:
        from overlapnet_amd.engine import QueryAhead
        qa = QueryAhead(eng, w, S.REFERENCE_MODEL_CFG)
        qa.submit(next_query())

    def step_warm():
        if qa is None:
            return step_warm_serial()
        qa.submit(next_query())
        fv, sp = qa.take()
        if spectral:
            return finish(eng.heads(cands, fv, spec_l=cand_spec, spec_r=sp, dcache_l=cand_dc))
        return finish(eng.heads(cands, fv))

    def make_step_cold(raw):
        def step_cold():
            if raw is not None:
                imgs_dev = eng.project(raw[0], raw[1], raw[2], want=(), stacked_flags=flags)["stacked"]
            else:
                imgs_dev = all_imgs
            eng.leg(imgs_dev, out=all_fv)
            cf, qf = all_fv[:P], all_fv[P:]
            if spectral:
                sp = eng.spectrum(all_fv)
                # (no Delta cache here: its rows pay when a candidate meets many queries; built and used once they only move work)
                return finish(eng.heads(cf, qf, spec_l=sp[:P], spec_r=sp[P:]))
            return finish(eng.heads(cf, qf))
        return step_cold

    raw = None
    if args.mode != "warm":
        all_fv = torch.empty((P + 1, 360, 128), dtype=torch.float32, device=dev)
        raw = make_raw() if args.mode == "fullstack" else None
        step = make_step_cold(raw)
    else:
        step = step_warm
    # dead-channel compaction (ovn_set_head_compaction): the contraction walks ceil(live / 32) of the 4 channel slices, where `live`
    # counts the channels that are non-zero somewhere in the QUERY's 360 columns -- measured here on the queries of the timed stream
    # (BEFORE the timed region -- the launches after it stay full-size sweeps, which the traffic passes read --
    #  read back from the library after an untimed sweep per query: ovn_head_walk_stats -- the slices every wave of the contraction kernel walks)
    live_counts, walks = [], []
    if args.mode == "warm" and args.head_precision == "f16x3" and spectral and P > 0 and not args.no_walk_stats:
        n_w = min(P, 32)
        for qi in query_ring:
            fq = eng.leg(qi)
            eng.heads(cands[:n_w], fq, spec_l=cand_spec[:n_w], spec_r=eng.spectrum(fq), dcache_l=cand_dc[:n_w] if cand_dc is not None else None)
            st = eng.head_walk_stats()
            live_counts.append(st["live_channels"])
            walks.append(st["k_walk_frac"])
    walk_frac = (sum(walks) / len(walks)) if walks else 1.0
    elapsed, prof, res = timed(step, args.warmup, args.steps, eng, use_dist, dev, side_eng=qa.side if qa is not None else None)
    if args.mode == "warm":
        # query 0 in a step of its own (untimed): what the accuracy block below and the same-results checks compare.  Serial order:
        # the same kernels and bits as the streamed order (tests/test_gpu_parity.py::test_query_ahead_*; `warm_serial.same_results`)
        res = step_warm_serial(query_img)
        streamed_same = None
        if qa is not None:      # ... and once more through the streamed path: must be the same bits
            qa.take()           # (the query in flight since the last timed step)
            qa.submit(query_img)
            fv_s, sp_s = qa.take()
            rs = finish(eng.heads(cands, fv_s, spec_l=cand_spec, spec_r=sp_s, dcache_l=cand_dc) if spectral else eng.heads(cands, fv_s))
            qa.submit(next_query())     # one query in flight again, as the latency sub-record expects
            if res is not None:
                streamed_same = bool(torch.equal(rs[0], res[0]) and torch.equal(rs[1], res[1]))
        torch.cuda.synchronize()

    # ---- census of the ranks (so that the first real multi-GPU line certifies itself): every rank reports the device it runs on;
    #      rank 0 puts backend, RCCL version, ranks seen and distinct devices into the line ----
    census = None
    if use_dist:
        pr = torch.cuda.get_device_properties(dev)
        mine_c = {"rank": rank, "local_rank": local_rank, "device": pr.name, "arch": getattr(pr, "gcnArchName", ""),
                  "pci_bus_id": getattr(pr, "pci_bus_id", None), "uuid": str(getattr(pr, "uuid", "")), "pid": os.getpid()}
        census = [None] * world
        dist.all_gather_object(census, mine_c)

    # ---- sharded pool: EVERY rank re-evaluates three windows of its block WITHOUT the Delta cache rows (same bits required; windows
    #      beyond the first 1024-pair chunk where the block is long enough) and rank 0 collects the verdicts ----
    shard_check = None
    if strong and args.mode == "warm" and spectral:
        own = eng.heads(cands, query_fv, spec_l=cand_spec, spec_r=query_spec, dcache_l=cand_dc)   # query 0 (left by the untimed step)
        wins = sorted({max(0, min(P - 64, s0)) for s0 in (0, 1024 + 37, P - 64)}) if P > 0 else []
        same = True
        for s0 in wins:
            n_w = min(64, P - s0)
            li = np.arange(s0, s0 + n_w, dtype=np.int32)      # an index list into the block: every candidate keeps its slot
            a = eng.heads(cands, query_fv, lidx=li, spec_l=cand_spec, spec_r=query_spec)
            same = same and bool(torch.equal(a["overlap"], own["overlap"][s0:s0 + n_w]) and torch.equal(a["yaw"], own["yaw"][s0:s0 + n_w]))
        mine = {"rank": rank, "block": [int(lo), int(hi)], "same": bool(same), "windows": [[int(s0), int(min(64, P - s0))] for s0 in wins]}
        if use_dist:
            shard_check = [None] * world
            dist.all_gather_object(shard_check, mine)
        else:
            shard_check = [mine]

    if rank != 0:
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()
        if qa is not None:
            qa.close()
        eng.close()
        return

    pairs = n_total * args.steps
    ms_step = 1e3 * elapsed / args.steps
    d_ms, d_n = prof["delta_c12"]
    avg_ms = d_ms / max(d_n, 1)
    launch_pairs = min(P, 2048)   # the heads run in chunks of <= 2048 pairs per launch
    if d_n:
        # launches of one step: ceil(P / 2048) of up to 2048 pairs; the average launch carries P * steps / d_n pairs
        launch_pairs = P * args.steps / d_n
    kname, peak, kprefix, flop_per_pair, rl_note = HEAD_KERNEL[args.head_precision]
    achieved = flop_per_pair * launch_pairs / (avg_ms * 1e-3) / 1e12 if d_n else 0.0
    mfma_per_alg = (3.0 * walk_frac) if args.head_precision == "f16x3" else 1.0
    if strong:
        workload = ("1-vs-%d synthetic candidate pool sharded over %d rank(s) in contiguous blocks (BASELINE configs[3]; warm: 1 query "
                    "leg per rank + %d head pairs per step in total), feature volumes generated on the device, 64x900x%d query"
                    % (n_total, world, n_total, C))
    else:
        workload = {"warm": "1-vs-%d candidate sweep per GPU (warm: 1 query leg + %d head pairs per step), 64x900x%d range images"
                            % (P, P, C),
                    "cold": "1-vs-%d sweep per GPU, COLD: %d legs from images in HBM + %d head pairs per step, 64x900x%d"
                            % (P, P + 1, P, C),
                    "fullstack": "1-vs-%d sweep per GPU from RAW scans: projection+normals of %d clouds, %d legs, %d head pairs per "
                                 "step, 64x900x%d" % (P, P + 1, P + 1, P, C)}[args.mode]
    out = {
        "metric": "scan-pairs/s (64x900 range images)", "value": pairs / elapsed, "unit": "pairs/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
        "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
        "dtype": DTYPE_LABEL[args.head_precision], "data": "synthetic",
        "config": {"workload": workload, "mode": args.mode, "pairs_per_step": n_total, "pairs_per_rank": P, "channels": C,
                   "weights": "seeded synthetic (no trained weights ship)", "head_precision": args.head_precision,
                   "leg_precision": leg_precision, "correlation_head": args.corr,
                   "delta_contraction": ("all 128 feature channels (ovn_set_head_compaction 0)" if args.no_compaction else
                                         "exact dead-channel compaction: channels that are zero in all columns of the query (of a column-group "
                                         "pair) are dropped from K -- roofline.k_walk_frac of the 128 walked, same results to fp32 rounding; "
                                         "dense_walk = the same step without it"),
                   "query_leg": ("step k + 1's query leg + spectrum on a second context / stream beside step k's head kernels "
                                 "(engine.QueryAhead; one leg and one head sweep enqueued per step)" if qa is not None
                                 else "on the heads' stream, in front of them"),
                   "collective": "RCCL gather of (overlap,yaw) to rank 0 per step" if use_dist else "none"},
        "roofline": {"bound": "mfma", "kernel": kname, "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                     "frac": achieved / peak, "traffic": None,
                     "flop_per_launch": flop_per_pair * launch_pairs, "pairs_per_launch": launch_pairs,
                     "avg_launch_ms": avg_ms,
                     "mfma_flops_per_algorithmic_flop": mfma_per_alg,
                     "frac_executed": mfma_per_alg * achieved / peak,
                     "query_live_channels": (sum(live_counts) / len(live_counts)) if live_counts else None,
                     "k_walk_frac": walk_frac,
                     "k_walk_by_query": {"live_channels": live_counts, "k_walk_frac": walks,
                                         "note": "the eight scans of the timed query stream; k_walk_frac = MFMAs the contraction kernel issues / those of "
                                                 "the 128-channel walk (a wave skips a 32-channel slice none of its three column "
                                                 "groups walks; ovn_head_walk_stats)"} if walks else None,
                     "delta_total_ms": sum(prof[k][0] / max(prof[k][1], 1) for k in ("delta_prep", "delta_c12", "delta_c2") if k in prof),
                     "note": rl_note + ("; the next query's leg kernels run on a second stream beside this kernel (QueryAhead): its event "
                                        "time includes the CUs they take at its round boundaries" if qa is not None else "")},
        "kernels": kernel_table(prof),
        "head_hbm_gbps_algorithmic": (pairs / elapsed) * CAND_BYTES_PER_PAIR / 1e9,
    }

    if census is not None:
        try:
            rccl = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:
            rccl = None
        devs = {(c["local_rank"], c["pci_bus_id"], c["uuid"]) for c in census}
        out["distributed"] = {"backend": dist.get_backend(), "rccl_version": rccl, "world_size": world,
                              "ranks_seen": sorted(c["rank"] for c in census), "distinct_devices": len(devs),
                              "one_rank_per_gpu": bool(len(devs) == world and not rehearsal),
                              "devices": [c["device"] for c in census], "archs": sorted({c["arch"] for c in census}),
                              "note": ("%d rank(s) seen over %s; " % (len(census), dist.get_backend())) +
                                      ("all ranks on ONE device (rehearsal)" if rehearsal else "%d distinct GPUs" % len(devs))}
        print("[bench] %s" % out["distributed"]["note"], "RCCL", rccl, file=sys.stderr, flush=True)
    if rehearsal:
        # ---- the gathered sweep of query 0 against ONE process evaluating the whole pool: must be the same bits ----
        pool_all = torch.empty((n_total, 360, 128), dtype=torch.float32, device=dev)
        for r_ in range(world):
            lo_r, hi_r = D.shard_bounds(pool_total, world, r_, D.SLOT_ALIGN)
            g_r = torch.Generator(device=dev).manual_seed(1234 + r_)
            for s_ in range(0, hi_r - lo_r, 4096):
                n_ = min(4096, hi_r - lo_r - s_)
                pool_all[lo_r + s_:lo_r + s_ + n_] = torch.relu(torch.randn((n_, 360, 128), device=dev, generator=g_r) + 0.1)
        full = eng.heads(pool_all, query_fv, spec_l=eng.spectrum(pool_all), spec_r=query_spec, dcache_l=eng.delta_cache(pool_all))
        out["rehearsal"] = True
        out["rehearsal_report"] = {
            "ranks": world, "devices": 1, "backend": dist.get_backend(),
            "gather_equals_single_process": bool(torch.equal(res[0].cpu(), full["overlap"].cpu())
                                                 and torch.equal(res[1].cpu().long(), full["yaw"].cpu().long())),
            "gathered": int(res[0].numel()), "pairs_per_s_all_ranks_sharing_one_gpu": out["value"],
            "note": "all ranks time-share ONE GPU over gloo / host collectives: the command path of the multi-GPU run, not a scaling "
                    "measurement; no N > 1 RCCL run exists"}
        out["value"] = None
        del pool_all, full

    # ---- accuracy part of the metric ("overlap MAE vs ref"), untimed ----
    ov = res[0].float().cpu().numpy()
    yw = res[1].cpu().numpy()
    golden_ok = (not strong and args.mode != "fullstack" and P == 1024 and C == 4 and world == 1
                 and os.path.isfile(os.path.join(ROOT, "tests", "golden", "parity_sweep_glorot.npz")))
    if golden_ok:
        out.update(golden_accuracy(ov[:P], yw[:P]))
    elif strong and shard_check is not None:
        # sharded pool (no committed oracle outputs): the verdicts of EVERY rank's self-check (above), and a live fp64 oracle on a few
        # pairs of rank 0's block
        fvq = query_fv                          # query 0, left there by the untimed step above
        wins = [w0 for w0, _ in shard_check[0]["windows"]]
        out["same_results_without_delta_cache"] = bool(all(c["same"] for c in shard_check))
        out["same_results_ranks"] = [bool(c["same"]) for c in shard_check]
        out["same_results_windows"] = shard_check[0]["windows"]
        out["shard_blocks"] = [c["block"] for c in shard_check]
        if args.accuracy_pairs > 0:
            from oracle import overlapnet_oracle as O
            k = min(args.accuracy_pairs, 4)
            s0 = wins[-1]
            fv64 = cands[s0:s0 + k].cpu().numpy().reshape(k, 1, 360, 128).astype(np.float64)
            q64 = np.repeat(fvq.cpu().numpy().reshape(1, 1, 360, 128).astype(np.float64), k, axis=0)
            o_ov, o_yaw, _, _ = O.heads_forward(fv64, q64, w)
            out["overlap_maxerr_vs_oracle"] = float(np.max(np.abs(ov[s0:s0 + k] - o_ov)))
            out["yaw_exact_rate"] = float(np.mean(yw[s0:s0 + k] == o_yaw))
            out["accuracy_pairs"] = int(k)
            out["accuracy_scope"] = "feature volumes -> heads vs fp64 oracle, pairs %d.. of rank 0's block" % s0
    elif args.accuracy_pairs > 0 and not strong:
        from oracle import overlapnet_oracle as O
        k = min(args.accuracy_pairs, P)
        if raw is not None:
            # fullstack: the step started from raw clouds -> the oracle starts from the same clouds (its own
            # projection + normals + channel stacking; NumPy's float32 angle functions restated, like the HIP kernel)
            offs = raw[1].cpu().numpy()
            rows = []
            for i in list(range(k)) + [P]:
                pts_i = raw[0][int(offs[i]):int(offs[i + 1])].cpu().numpy()
                rng_i, vtx_i, itn_i, _ = O.range_projection(pts_i)
                rows.append(S.stack(rng_i, O.gen_normal_map(rng_i, vtx_i), itn_i, flags))
            acc_in = np.stack(rows)
            out["accuracy_scope"] = "raw clouds -> projection -> leg -> heads vs fp64 oracle (own projection)"
        else:
            first = next(S.sweep_pool_images(P, C, rank, fx))[1][:k]
            acc_in = np.concatenate([first, query_img.cpu().numpy()], axis=0)
            out["accuracy_scope"] = "images -> leg -> heads vs fp64 oracle"
        ofv = O.leg_forward(acc_in, w, S.REFERENCE_MODEL_CFG, np.float64)
        o_ov, o_yaw, _, _ = O.heads_forward(ofv[:k], np.repeat(ofv[k:k + 1], k, axis=0), w)
        out["overlap_mae_vs_oracle"] = float(np.mean(np.abs(ov[:k] - o_ov)))
        out["overlap_maxerr_vs_oracle"] = float(np.max(np.abs(ov[:k] - o_ov)))
        out["yaw_exact_rate"] = float(np.mean(yw[:k] == o_yaw))
        out["accuracy_pairs"] = int(k)

    # ---- sub-records: every other number DESIGN.md quotes, measured in this same run (N = 1 only) ----
    if world == 1 and not strong and not args.no_extras and args.mode == "warm":
        sub_steps = max(args.steps, 20)
        # (0) the same warm step with the query leg IN FRONT of its heads on one stream (no QueryAhead)
        if qa is not None:
            e0, p0, r0 = timed(step_warm_serial, 2, sub_steps, eng, False, dev)
            out["warm_serial"] = {"value": P * sub_steps / e0, "unit": "pairs/s", "ms_per_step": 1e3 * e0 / sub_steps, "steps": sub_steps,
                                  "step": "1 query leg, then %d head pairs, one stream" % P,
                                  "same_results": streamed_same}     # query 0 through both orders, after the timed region
        # (0b) the same streamed step walking all 128 channels (no dead-channel compaction): what a query without dead channels costs
        if not args.no_compaction:
            eng.set_head_compaction(False)
            try:
                e0b, p0b, r0b = timed(step_warm, 2, sub_steps, eng, False, dev, side_eng=qa.side if qa is not None else None)
            finally:
                eng.set_head_compaction(True)
            c12_dense = p0b["delta_c12"][0] / max(p0b["delta_c12"][1], 1)
            out["dense_walk"] = {"value": P * sub_steps / e0b, "unit": "pairs/s", "ms_per_step": 1e3 * e0b / sub_steps, "steps": sub_steps,
                                 "delta_c12_ms": c12_dense,
                                 "frac": flop_per_pair * P / (c12_dense * 1e-3) / 1e12 / peak,
                                 "step": "the timed step with ovn_set_head_compaction(0): every pair walks all 128 feature channels"}
        # (0c) the SECOND weight set (trained-like dynamic range, tools/synthetic.make_trained_like_weights: the set the parity sweep
        #      checks pair by pair) through the same streamed step: its own engine, pool features, spectra, Delta rows and query-ahead
        #      context; compaction on, then off.  What the compaction buys depends on the query's dead channels, i.e. on the weights.
        if args.head_precision == "f16x3" and spectral and pool_imgs is not None and not args.no_compaction:
            out["trained_like"] = trained_like_record(C, P, dev, pool_imgs, query_ring, query_img, sub_steps, flop_per_pair, peak)
        # (1) everything on the fp32 matrix cores, direct correlation form
        eng.set_head_precision("f32")
        eng.set_leg_precision("f32")

        def step_f32():
            eng.leg(query_img, out=query_fv)
            r = eng.heads(cands_f32, query_fv)
            return r["overlap"], r["yaw"]
        cands_f32 = torch.empty_like(cands)
        for s in range(0, P, 128):
            eng.leg(pool_imgs[s:s + 128], out=cands_f32[s:s + 128])
        e2, p2, r2 = timed(step_f32, 2, 5, eng, False, dev)
        rec = {"value": P * 5 / e2, "unit": "pairs/s", "ms_per_step": 1e3 * e2 / 5, "steps": 5,
               "arithmetic": "leg + Delta head + direct correlation head on v_mfma_f32_16x16x4_f32 (bit-for-bit an fp32 FMA chain)",
               "delta_c12_ms": p2["delta_c12"][0] / max(p2["delta_c12"][1], 1)}
        if golden_ok:
            ga = golden_accuracy(r2[0].float().cpu().numpy(), r2[1].cpu().numpy())
            rec.update({k: ga[k] for k in ("overlap_mae_vs_oracle", "overlap_maxerr_vs_oracle", "yaw_exact_rate", "accuracy_pairs")})
        out["fp32_mode"] = rec
        del cands_f32
        eng.set_head_precision(args.head_precision)
        eng.set_leg_precision(leg_precision)
        # (2) cold: the legs of all candidates inside the step
        all_fv = torch.empty((P + 1, 360, 128), dtype=torch.float32, device=dev)
        e3, p3, r3 = timed(make_step_cold(None), 2, sub_steps, eng, False, dev)
        leg_ms = p3["leg_conv"][0] / sub_steps
        out["cold"] = {"value": P * sub_steps / e3, "unit": "pairs/s", "ms_per_step": 1e3 * e3 / sub_steps, "steps": sub_steps,
                       "step": "%d legs from images in HBM + spectra + %d head pairs" % (P + 1, P), "leg_ms_per_step": leg_ms,
                       "leg_scans_per_s": (P + 1) / (leg_ms * 1e-3),
                       "leg_frac_of_16bit_mfma_peak_algorithmic": (P + 1) * LEG_FLOP_PER_SCAN.get(C, 1733.2e6) / (leg_ms * 1e-3) / 1e12
                       / PEAK_16BIT_MFMA_TFLOPS}
        # (3) fullstack: raw clouds -> projection + normals -> legs -> heads
        raw2 = make_raw()
        e4, p4, r4 = timed(make_step_cold(raw2), 2, sub_steps, eng, False, dev)
        out["fullstack"] = {"value": P * sub_steps / e4, "unit": "pairs/s", "ms_per_step": 1e3 * e4 / sub_steps, "steps": sub_steps,
                            "step": "projection + normals of %d raw clouds, %d legs, spectra, %d head pairs" % (P + 1, P + 1, P),
                            "projection_ms_per_step": p4["projection"][0] / sub_steps,
                            "projection_scans_per_s": (P + 1) / (p4["projection"][0] / sub_steps * 1e-3)}
        # per scan 16 B per point in + 64 x 900 x (1 + 3 + 1) x 4 B of images out
        proj_bytes = 16.0 * float(raw2[1][-1].item()) + (P + 1) * 64 * 900 * 5 * 4
        out["fullstack"]["projection_frac_of_hbm_peak"] = proj_bytes / (p4["projection"][0] / sub_steps * 1e-3) / PEAK_HBM_BPS
        fs_golden = os.path.join(ROOT, "tests", "golden", "parity_fullstack.npz")
        if P == 1024 and C == 4 and os.path.isfile(fs_golden):
            # the timed fullstack step's own results against the fp64 oracle that started from the same raw clouds (committed
            # outputs, tests/golden/make_fullstack_golden.py: restated projection + normals + fp64 leg + heads), first 64 pairs
            with np.load(fs_golden) as z:
                g_ov, g_yaw = z["overlap"], z["yaw"]
            k = len(g_ov)
            f_ov, f_yaw = r4[0][:k].float().cpu().numpy(), r4[1][:k].cpu().numpy()
            out["fullstack"].update({"accuracy_pairs": int(k), "overlap_maxerr_vs_oracle": float(np.max(np.abs(f_ov - g_ov))),
                                     "overlap_mae_vs_oracle": float(np.mean(np.abs(f_ov - g_ov))),
                                     "yaw_exact_rate": float(np.mean(f_yaw == g_yaw)),
                                     "accuracy_scope": "raw clouds -> projection -> leg -> heads on the GPU vs the fp64 oracle from the same "
                                                       "clouds (tests/golden/parity_fullstack.npz)"})
        del raw2, all_fv
        # (4) the correlation head alone (the kernel the north star puts an HBM-roofline number on), N = 1024 and 16384
        ch = {}
        for n_c in (1024, 16384):
            gen = torch.Generator(device=dev).manual_seed(99)
            f = cands if n_c == P else torch.relu(torch.randn((n_c, 360, 128), device=dev, generator=gen) + 0.1)
            sp = cand_spec if n_c == P else eng.spectrum(f)
            qs = eng.spectrum(query_fv)
            e5, p5, _ = timed(lambda: eng.corr_head_spectral(sp, qs), 3, sub_steps, eng, False, dev)
            ms = p5["corr_spectral"][0] / p5["corr_spectral"][1]
            ch["n%d" % n_c] = {"ms": ms, "pairs_per_s": n_c / (ms * 1e-3),
                               "algorithmic_bytes": n_c * CAND_BYTES_PER_PAIR, "streamed_bytes": n_c * SPEC_BYTES_PER_PAIR,
                               "frac_of_8TBps": n_c * CAND_BYTES_PER_PAIR / (ms * 1e-3) / PEAK_HBM_BPS,
                               "frac_of_8TBps_streamed": n_c * SPEC_BYTES_PER_PAIR / (ms * 1e-3) / PEAK_HBM_BPS}
            del f, sp
        if spectral and prof.get("corr_spectral", (0, 0))[1]:
            ms_in = prof["corr_spectral"][0] / prof["corr_spectral"][1]
            ch["in_step_n%d" % P] = {"ms": ms_in, "frac_of_8TBps": P * CAND_BYTES_PER_PAIR / (ms_in * 1e-3) / PEAK_HBM_BPS,
                                     "note": "the yaw kernel inside the timed warm step (between the Delta kernels' 8.9 GB of traffic: "
                                             "nothing of the candidate spectra survives in the Infinity Cache from one step to the next)"}
        ch["n1024"]["infinity_cache_resident"] = True   # 193 MB of spectra re-read by back-to-back launches: not an HBM figure
        ch["note"] = ("spectral form on cached candidate spectra; `frac_of_8TBps` prices SURVEY.md 8d's algorithmic bytes (184,328 B per "
                      "pair), `..._streamed` the 188,420 B the kernel actually reads and writes per pair; ms = HIP events around the launch(es)")
        out["corr_head"] = ch
        # (5) BASELINE configs[2]: the loop-closure sweep of a whole sequence through the drop-in API -- 1101 frames (KITTI 07's length,
        #     emulated with synthetic scans), frame i against ALL i cached frames (605,550 pairs) via
        #     `Infer.infer_multiple` (files on disk -> results on the host), next to the same sweep at engine level
        #     (device-resident inputs, decision on the device), tools/bench_infer_api.py
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from bench_infer_api import api_sweep, engine_sweep
        fr = 1101
        es, aps = engine_sweep(fr, C), api_sweep(fr, "multiple")
        out["infer_api"] = {"frames": fr, "pairs": aps["pairs"], "api_frames_per_s": aps["frames_per_s"], "api_pairs_per_s": aps["pairs_per_s"],
                            "api_seconds": aps["seconds"], "engine_frames_per_s": es["frames_per_s"], "engine_pairs_per_s": es["pairs_per_s"],
                            "engine_seconds": es["seconds"], "api_over_engine": aps["frames_per_s"] / es["frames_per_s"],
                            "workload": "BASELINE configs[2] (1-vs-all-previous over 1101 frames, ungated)",
                            "step": "frame i: np.load depth+normal .npy, H2D, leg, spectrum, both heads vs ALL i cached frames, results to host"}
        # (6) BASELINE configs[0] and small gated sweeps: latency of ONE query against N cached candidates -- query leg + spectrum +
        #     both heads + the on-device decision, the 16-byte record read back by the host every query (demo2: N = 1; demo3 after
        #     gating: tens of candidates)
        from overlapnet_amd.engine import decode_match
        lat = {}
        for n_c in (1, 16, 100, 256):
            def q_step():
                eng.leg(query_img, out=query_fv)
                eng.spectrum(query_fv, out=query_spec)
                r = eng.heads(cands[:n_c], query_fv, spec_l=cand_spec[:n_c], spec_r=query_spec,
                              dcache_l=cand_dc[:n_c] if cand_dc is not None else None)
                return decode_match(eng.best_match(r["overlap"], r["yaw"], 0.3, host=True))   # record written to pinned host memory
            for _ in range(5):
                q_step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(50):
                q_step()
            torch.cuda.synchronize()
            lat["n%d" % n_c] = {"ms_per_query": 1e3 * (time.perf_counter() - t0) / 50}
            if qa is not None:
                # the same queries as a STREAM (a recorded sequence, demo3_lcd.py:88-123): query k + 1's leg runs on the second context
                # beside query k's heads; the host still reads every decision before it enqueues the next heads
                def q_stream():
                    fv, sp = qa.take()
                    r = eng.heads(cands[:n_c], fv, spec_l=cand_spec[:n_c], spec_r=sp,
                                  dcache_l=cand_dc[:n_c] if cand_dc is not None else None)
                    rec = eng.best_match(r["overlap"], r["yaw"], 0.3)
                    qa.submit(query_img, wait_current=False)     # the next query's image is resident: its leg need not wait for these heads
                    return decode_match(rec)                     # (device record: the copy overlaps the submit above)
                for _ in range(5):
                    q_stream()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(50):
                    q_stream()
                torch.cuda.synchronize()
                lat["n%d" % n_c]["ms_per_query_streamed"] = 1e3 * (time.perf_counter() - t0) / 50
        lat["note"] = ("host wall clock per query incl. the device-to-host read of the decision; N = 1 is BASELINE configs[0] (demo2's single "
                       "pair); ms_per_query = one isolated query (leg, spectrum, heads, decision back to back); ms_per_query_streamed = per "
                       "query of a stream whose next leg runs beside the current heads (QueryAhead), decision still read every query")
        out["latency"] = lat
    # ---- HBM traffic of the dominant kernel: measured now (two rocprofv3 PMC passes over a short run of this program) when this is the
    #      default single-GPU warm configuration, else the newest committed profile; never silently ----
    traffic_mode = args.traffic if (world == 1 and not strong and args.mode == "warm" and not args.no_extras) else \
        ("none" if args.traffic == "none" else "committed")
    pass_args = ["--pool", str(P), "--channels", str(C), "--head-precision", args.head_precision, "--corr", args.corr] + \
        (["--no-compaction"] if args.no_compaction else [])
    t_bytes, t_src = rocprof_traffic(kprefix, traffic_mode, pass_args)
    # the scalars DESIGN.md quotes, copied into `roofline`: the driver's record keeps the first ~24 keys of that object (names cut at 40
    # characters, strings at 120), sub-records survive only as key names -- so the object is built in ORDER OF IMPORTANCE; all measured
    # in THIS run
    rl0 = out["roofline"]
    rl0["traffic"] = t_bytes
    rl = {k: rl0[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac")}

    def sub(name, key, as_name):
        if name in out and key in out[name]:
            rl[as_name] = out[name][key]
    # `frac` counts the ALGORITHMIC flops of a sweep whose K walk skipped the query's dead channels (k_walk_frac of the 128 walked);
    # `frac_dense` is the same kernel walking all 128 -- what ANY query is guaranteed.  Both, for both weight sets.
    sub("dense_walk", "frac", "frac_dense")
    for k in ("traffic", "frac_executed", "k_walk_frac"):
        rl[k] = rl0[k]
    sub("dense_walk", "value", "dense_walk_pairs_per_s")
    if "trained_like" in out:
        tl = out["trained_like"]
        rl["value_trained_like"] = tl["compacted"]["value"]
        rl["frac_trained_like"] = tl["compacted"]["frac"]
        rl["frac_dense_trained_like"] = tl["dense_walk"]["frac"]
        rl["k_walk_frac_trained_like"] = tl["k_walk_frac"]

    sub("cold", "value", "cold_pairs_per_s")
    sub("cold", "leg_scans_per_s", "cold_leg_scans_per_s")
    sub("cold", "leg_frac_of_16bit_mfma_peak_algorithmic", "cold_leg_frac_of_mfma_peak")
    sub("fullstack", "value", "fullstack_pairs_per_s")
    sub("fullstack", "projection_scans_per_s", "fullstack_projection_scans_per_s")
    sub("fullstack", "projection_frac_of_hbm_peak", "projection_frac_of_hbm_peak")
    if "overlap_maxerr_vs_oracle" in out:
        rl["overlap_maxerr_vs_oracle"] = out["overlap_maxerr_vs_oracle"]
    if prof.get("corr_spectral", (0, 0))[1] and spectral:
        ms_in = prof["corr_spectral"][0] / prof["corr_spectral"][1]
        rl["corr_in_step_frac_of_hbm_peak"] = P * CAND_BYTES_PER_PAIR / (ms_in * 1e-3) / PEAK_HBM_BPS
    if "latency" in out:
        rec = out["latency"].get("n1", {})
        if "ms_per_query" in rec:
            rl["latency_n1_ms"] = rec["ms_per_query"]
    # ---- (beyond the driver's 24 keys: kept in the line itself) ----
    if "latency" in out:
        rec = out["latency"].get("n1", {})
        if "ms_per_query_streamed" in rec:
            rl["latency_n1_streamed_ms"] = rec["ms_per_query_streamed"]
    rl["step_pairs_per_s"] = out["value"]
    sub("fullstack", "overlap_maxerr_vs_oracle", "fullstack_overlap_maxerr_vs_oracle")
    if "corr_head" in out and "n16384" in out["corr_head"]:
        rl["corr_n16384_frac_of_hbm_peak"] = out["corr_head"]["n16384"]["frac_of_8TBps"]
    if "trained_like" in out:
        rl["dense_walk_trained_like_pairs_per_s"] = out["trained_like"]["dense_walk"]["value"]
        if "overlap_maxerr_vs_oracle" in out["trained_like"]:
            rl["overlap_maxerr_vs_oracle_trained_like"] = out["trained_like"]["overlap_maxerr_vs_oracle"]
    rl["traffic_from"] = (t_src.get("source") or "none") if isinstance(t_src, dict) else str(t_src)
    if args.head_precision == "f16x3":
        # executed MFMA rate against what a bare MFMA loop sustains on real operands (a committed measurement, not this run's)
        rl["executed_frac_of_sustained_mfma_rate"] = mfma_per_alg * rl0["achieved"] / SUSTAINED_16BIT_MFMA_TFLOPS
    rl["avg_launch_ms"] = rl0["avg_launch_ms"]
    sub("fp32_mode", "value", "fp32_mode_pairs_per_s")
    sub("warm_serial", "value", "warm_serial_pairs_per_s")
    if "yaw_exact_rate" in out:
        rl["yaw_exact_rate"] = out["yaw_exact_rate"]
    if "infer_api" in out:
        rl["infer_api_frames_per_s"] = out["infer_api"]["api_frames_per_s"]
        rl["infer_api_over_engine"] = out["infer_api"]["api_over_engine"]
    sub("fp32_mode", "overlap_maxerr_vs_oracle", "fp32_mode_overlap_maxerr_vs_oracle")
    sub("fullstack", "yaw_exact_rate", "fullstack_yaw_exact_rate")
    if "latency" in out:
        rec = out["latency"].get("n100", {})
        if "ms_per_query" in rec:
            rl["latency_n100_ms"] = rec["ms_per_query"]
        if "ms_per_query_streamed" in rec:
            rl["latency_n100_streamed_ms"] = rec["ms_per_query_streamed"]
    if prof.get("corr_spectral", (0, 0))[1] and spectral:
        rl["corr_in_step_ms"] = prof["corr_spectral"][0] / prof["corr_spectral"][1]
    if "accuracy_pairs" in out:
        rl["accuracy_pairs"] = out["accuracy_pairs"]
    if isinstance(t_src, dict):
        rl["traffic_detail"] = t_src
    for k, v in rl0.items():            # flop_per_launch, pairs_per_launch, delta_total_ms, note, ...
        rl.setdefault(k, v)
    out["roofline"] = rl
    # next to `value` at the top level: what the same step does without the data-dependent part, and on the second weight set
    if "dense_walk" in out:
        out["value_dense_walk"] = out["dense_walk"]["value"]
    if "trained_like" in out:
        out["value_trained_like"] = out["trained_like"]["compacted"]["value"]
        out["value_trained_like_dense_walk"] = out["trained_like"]["dense_walk"]["value"]
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(C, P)
        if not args.no_extras:
            # the preprocessing half of BASELINE configs[4] on the same host, next to `fullstack`'s projection
            out["cpu_baseline_preprocess"] = cpu_baseline_preprocess()
            out["cpu_baseline"]["preprocess_scans_per_s"] = out["cpu_baseline_preprocess"]["value"]
            out["cpu_baseline"]["preprocess_kind"] = out["cpu_baseline_preprocess"]["kind"]
            out["roofline"]["cpu_preprocess_scans_per_s"] = out["cpu_baseline_preprocess"]["value"]
    print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if qa is not None:
        qa.close()
    eng.close()


if __name__ == "__main__":
    main()
