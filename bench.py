#!/usr/bin/env python3
"""Benchmark of the OverlapNet hot path on MI355X -- BASELINE.json metric: scan-pairs/s on 64x900 range images.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--pool P] [--channels C]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One STEP = one loop-closure query the way the reference's `Infer.infer_multiple` runs it
(src/two_heads/infer.py:162-203): the leg over the query scan (1 scan, 64x900xC, already in HBM) plus BOTH
heads of that query against the P candidate feature volumes cached in HBM ("warm" sweep: candidates' legs ran
when they were the current frame, infer.py:184-185).  N = 1, P = 1024, C = 4 is BASELINE.json configs[1]
("batched 1-vs-1024 pairs, 64x900 depth+normals").  With N > 1 every rank holds its own P candidates
(weak scaling, 1-vs-N*P), the only collective is the per-step gather of (overlap, yaw) to rank 0.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from overlapnet_amd import distributed as D  # noqa: E402
from overlapnet_amd import synthetic as S  # noqa: E402
from overlapnet_amd.engine import OvnEngine  # noqa: E402

# algorithmic work of the dominant kernel (fused DeltaLayer + c_conv1 + c_conv2), SURVEY.md section 8a row a7:
#   c_conv1 8640 x 1920 x 64 and c_conv2 576 x 960 x 128 multiply-adds per pair, FLOP = 2 * MAC
DELTA_C12_FLOP_PER_PAIR = 2 * (8640 * 1920 * 64 + 576 * 960 * 128)
HEAD_FLOP_PER_PAIR = 2_550_646_784          # whole Delta head
CORR_FLOP_PER_PAIR = 33_177_600
CAND_BYTES_PER_PAIR = 184_320 + 8           # candidate feature volume read once + (overlap, yaw) written
PEAK_F32_MFMA_TFLOPS = 157.3                # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0              # MI355X_MICROARCH.md: dense bf16 MFMA peak (not the 2:1-sparse figure)


def rocprof_traffic(kernel_prefix: str):
    """HBM bytes per launch of the kernel whose name starts with `kernel_prefix`, from the NEWEST committed rocprofv3
    PMC summary under profiles/ (separate FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE doubled per MI355X_MICROARCH.md;
    see tools/summarize_rocprof.py) -- null if no summary names it.  Counters cannot be read live in-process."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_rocprof_summary.json")))  # r1 < r1b < ... by name
    for f in reversed(files):
        try:
            t = json.load(open(f)).get("hbm_traffic_per_launch", {})
        except Exception:
            continue
        hits = [v["total_bytes"] for k, v in t.items() if k.startswith(kernel_prefix)]
        if hits:
            return max(hits)
    return None


def cpu_baseline(channels: int, pool: int):
    """The CPU restatement oracle (PyTorch-CPU fp32, structured like the reference: leg model, then head
    model in batches of 16 with the 360x360x128 Delta tensor materialised) timed on this host's cores on
    a bounded sample: 1 leg + 32 pairs, extrapolated to the 1-leg + `pool`-pairs step."""
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pool", type=int, default=1024, help="candidate feature volumes resident per GPU")
    ap.add_argument("--channels", type=int, default=4, help="4 = depth+normals (network.yml), 1 = depth, 5 = +intensity")
    ap.add_argument("--head-precision", default="bf16x3", choices=["f32", "bf16x3"],
                    help="Delta-head contraction arithmetic: fp32 MFMA, or 3-term bf16 split on the bf16 MFMA (default)")
    ap.add_argument("--mode", default="warm", choices=["warm", "cold", "fullstack"],
                    help="warm (default, the BASELINE metric): candidate features cached, step = query leg + heads; "
                         "cold: step also runs the legs of all candidates from images resident in HBM; "
                         "fullstack: step starts from raw point clouds (projection + normals on the GPU)")
    ap.add_argument("--leg-precision", default="bf16x3", choices=["f32", "bf16x3"],
                    help="leg convolution arithmetic (the Infer class defaults to f32; both are parity-tested)")
    ap.add_argument("--corr", default="spectral", choices=["spectral", "direct"],
                    help="correlation head: spectral form on cached candidate spectra (default) or direct Gram form")
    ap.add_argument("--force-dist", action="store_true", help="initialise RCCL and run the gather even with one rank (path check)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--accuracy-pairs", type=int, default=12, help="pairs checked against the fp64 oracle (untimed)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    C, P = args.channels, args.pool
    eng = OvnEngine(64, 900, C, device=local_rank)
    w = S.make_test_weights(C, seed=0)
    eng.load_weights(w, S.REFERENCE_MODEL_CFG)
    eng.set_head_precision(args.head_precision)
    eng.set_leg_precision(args.leg_precision)

    # ---- untimed setup: candidate pool -> feature volumes resident in HBM (each rank its own pool) ----
    fx = S.load_fixture_images()
    cands = torch.empty((P, 360, 128), dtype=torch.float32, device=dev)
    cold_imgs = torch.empty((P, 64, 900, C), dtype=torch.float32, device=dev) if args.mode == "cold" else None
    chunk = 128
    acc_imgs = None  # host copy of the first candidates' images for the untimed end-to-end accuracy check
    for s in range(0, P, chunk):
        n = min(chunk, P - s)
        imgs = S.candidate_images(n, C, seed=1234 + 7919 * rank + s, fixture=fx)
        # distinct shifts across chunks / ranks
        imgs = np.ascontiguousarray(np.roll(imgs, (s * 37 + rank * 11) % 900, axis=2))
        if s == 0:
            acc_imgs = imgs[:max(1, min(args.accuracy_pairs, n))].copy()
        timg = torch.from_numpy(imgs).to(dev)
        if cold_imgs is not None:
            cold_imgs[s:s + n].copy_(timg)
        eng.leg(timg, out=cands[s:s + n])
    query_img = torch.from_numpy(S.stack(fx["range_0"], fx["normal_0"], fx["intensity_0"], S.flags_of(C))[None]).to(dev)
    query_fv = torch.empty((1, 360, 128), dtype=torch.float32, device=dev)
    raw = None
    if args.mode == "fullstack":
        # raw scans resident in HBM: the two shipped KITTI scans rotated about z by i * 360/P degrees
        base = [torch.from_numpy(fx["points_%d" % i]).to(dev) for i in range(2)]
        pts, offs = [], [0]
        for i in range(P + 1):
            b = base[i % 2]
            th = 2.0 * np.pi * ((i * 37) % 900) / 900.0
            c_, s_ = float(np.cos(th)), float(np.sin(th))
            q = b.clone()
            q[:, 0] = c_ * b[:, 0] - s_ * b[:, 1]
            q[:, 1] = s_ * b[:, 0] + c_ * b[:, 1]
            pts.append(q)
            offs.append(offs[-1] + q.shape[0])
        raw = (torch.cat(pts).contiguous(), torch.tensor(offs, dtype=torch.int64, device=dev), max(p.shape[0] for p in base))
        del pts
    flags = S.flags_of(C)
    all_fv = torch.empty((P + 1, 360, 128), dtype=torch.float32, device=dev) if args.mode != "warm" else None
    spectral = args.corr == "spectral"
    cand_spec = eng.spectrum(cands) if spectral else None          # cached per candidate, like its feature volume
    query_spec = torch.empty((1, 128, eng.SPEC_W), dtype=torch.float32, device=dev) if spectral else None
    torch.cuda.synchronize()

    def step_cold():
        if raw is not None:
            imgs_dev = eng.project(raw[0], raw[1], raw[2], want=(), stacked_flags=flags)["stacked"]
        else:
            imgs_dev = torch.cat([cold_imgs, query_img])
        eng.leg(imgs_dev, out=all_fv)
        cf, qf = all_fv[:P], all_fv[P:]
        if spectral:
            sp = eng.spectrum(all_fv)
            r = eng.heads(cf, qf, spec_l=sp[:P], spec_r=sp[P:])
        else:
            r = eng.heads(cf, qf)
        if use_dist:
            return D.gather_scores(r["overlap"], r["yaw"], P * world)
        return r["overlap"], r["yaw"]

    def step_warm():
        eng.leg(query_img, out=query_fv)
        if spectral:
            eng.spectrum(query_fv, out=query_spec)
            r = eng.heads(cands, query_fv, spec_l=cand_spec, spec_r=query_spec)
        else:
            r = eng.heads(cands, query_fv)
        if use_dist:
            return D.gather_scores(r["overlap"], r["yaw"], P * world)
        return r["overlap"], r["yaw"]

    step = step_warm if args.mode == "warm" else step_cold
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    eng.profile_begin()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    prof = eng.profile_end()
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        pairs = P * world * args.steps
        ms_step = 1e3 * elapsed / args.steps
        d_ms, d_n = prof["delta_c12"]
        avg_ms = d_ms / max(d_n, 1)
        achieved = DELTA_C12_FLOP_PER_PAIR * P / (avg_ms * 1e-3) / 1e12 if d_n else 0.0
        if args.head_precision == "f32":
            peak, kname, dtype = PEAK_F32_MFMA_TFLOPS, "delta_c12_kernel (DeltaLayer+c_conv1+c_conv2, fp32 MFMA)", "f32"
            rl_note = "fp32 matrix cores, one MFMA per product"
        else:
            peak, kname, dtype = PEAK_BF16_MFMA_TFLOPS, "delta_c12_bf16x3_j2_kernel (DeltaLayer+c_conv1+c_conv2, bf16 MFMA)", "bf16x3"
            rl_note = ("achieved counts ALGORITHMIC flops; the 3-term bf16 split issues 3 MFMA flops per algorithmic "
                       "flop, so the matrix pipe executes 3x this rate (frac <= 1/3 by construction)")
        kernels = {k: {"ms_per_launch": (v[0] / v[1] if v[1] else None), "launches": v[1]} for k, v in prof.items() if v[1]}
        # fp32 storage and accumulation everywhere; "bf16x3" = every fp32 product as three bf16 MFMA products (x = hi + lo),
        # ~2^-17 relative per product -- the accuracy fields below are measured in this very run against the fp64 oracle
        dtype_label = "f32" if dtype == "f32" else "f32 (3-term bf16 split on the bf16 MFMA, fp32 accumulate)"
        out = {
            "metric": "scan-pairs/s (64x900 range images)", "value": pairs / elapsed, "unit": "pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype_label, "data": "synthetic",
            "config": {"workload": {"warm": "1-vs-%d candidate sweep per GPU (warm: 1 query leg + %d head pairs per step), "
                                            "64x900x%d range images" % (P, P, C),
                                    "cold": "1-vs-%d sweep per GPU, COLD: %d legs from images in HBM + %d head pairs per step, "
                                            "64x900x%d" % (P, P + 1, P, C),
                                    "fullstack": "1-vs-%d sweep per GPU from RAW scans: projection+normals of %d clouds, %d legs, "
                                                 "%d head pairs per step, 64x900x%d" % (P, P + 1, P + 1, P, C)}[args.mode],
                       "mode": args.mode,
                       "pairs_per_step": P * world, "channels": C, "weights": "seeded synthetic (no trained weights ship)",
                       "correlation_head": args.corr,
                       "collective": "RCCL gather of (overlap,yaw) to rank 0 per step" if use_dist else "none"},
            "roofline": {"bound": "mfma", "kernel": kname, "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": rocprof_traffic("delta_c12_bf16x3" if dtype == "bf16x3" else "delta_c12_kernel"),
                         "flop_per_launch": DELTA_C12_FLOP_PER_PAIR * P, "avg_launch_ms": avg_ms, "note": rl_note},
            "kernels": kernels,
            "head_hbm_gbps_algorithmic": (P * world * args.steps / elapsed) * CAND_BYTES_PER_PAIR / 1e9,
        }
        # accuracy part of the metric ("overlap MAE vs ref"): untimed check against the fp64 oracle
        if args.accuracy_pairs > 0:
            from oracle import overlapnet_oracle as O
            k = min(args.accuracy_pairs, acc_imgs.shape[0])
            ov = res[0][:k].float().cpu().numpy()
            yw = res[1][:k].cpu().numpy()
            # end to end: fp64 oracle leg on the same images, then fp64 heads (l = candidate, r = query)
            if raw is not None:
                # fullstack: the step started from raw clouds -> the oracle starts from the same clouds (its own
                # projection + normals + channel stacking; fp64 trig rounded to fp32 like the HIP kernel)
                offs = raw[1].cpu().numpy()
                sel = list(range(k)) + [P]
                rows = []
                for i in sel:
                    pts_i = raw[0][int(offs[i]):int(offs[i + 1])].cpu().numpy()
                    rng_i, vtx_i, itn_i, _ = O.range_projection(pts_i, trig64=True)
                    nrm_i = O.gen_normal_map(rng_i, vtx_i)
                    rows.append(S.stack(rng_i, nrm_i, itn_i, flags))
                all_imgs = np.stack(rows)
                out["accuracy_scope"] = "raw clouds -> projection -> leg -> heads vs fp64 oracle (own projection)"
            else:
                all_imgs = np.concatenate([acc_imgs[:k], query_img.cpu().numpy()], axis=0)
            ofv = O.leg_forward(all_imgs, w, S.REFERENCE_MODEL_CFG, np.float64)
            fl = ofv[:k]
            fr = np.repeat(ofv[k:k + 1], k, axis=0)
            o_ov, o_yaw, _, _ = O.heads_forward(fl, fr, w)
            out.setdefault("accuracy_scope", "images -> leg -> heads vs fp64 oracle")
            out["overlap_mae_vs_oracle"] = float(np.mean(np.abs(ov - o_ov)))
            out["overlap_maxerr_vs_oracle"] = float(np.max(np.abs(ov - o_ov)))
            out["yaw_exact_rate"] = float(np.mean(yw == o_yaw))
            out["accuracy_pairs"] = int(k)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(C, P)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
