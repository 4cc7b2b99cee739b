#!/usr/bin/env python3
"""Streaming loop-closure sweep, engine level and THROUGH THE DROP-IN API (what demo3 sees after the documented import swap).

BASELINE.json configs[2] emulated (KITTI 07 is not in the tree): F synthetic frames; frame i goes through the leg
and is compared with ALL previous frames (ungated: F (F-1) / 2 pairs), feature / spectrum caches resident in HBM.
  * engine_sweep: device-resident input images, `OvnEngine` calls (next frame's leg beside the current frame's heads: `QueryAhead`),
                  decision on the device (one 16-byte record per frame);
  * api_sweep:    depth / normal .npy files laid out like demo1 writes them, `Infer.infer_multiple(i, [0 .. i-1])` (or
                  `infer_best_match`) per frame: np.load, H2D, leg, spectrum, both heads, result back on the host.
    python tools/bench_infer_api.py [--frames 1101] [--mode multiple|best_match]
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import synthetic as S  # noqa: E402


def engine_sweep(frames: int = 1101, C: int = 4):
    from overlapnet_amd.engine import OvnEngine, QueryAhead, decode_match
    eng = OvnEngine(64, 900, C)
    w = S.make_test_weights(C, 0)
    eng.load_weights(w, S.REFERENCE_MODEL_CFG)
    qa = QueryAhead(eng, w, S.REFERENCE_MODEL_CFG)     # frame i + 1's leg beside frame i's head kernels (what Infer does too)
    dev = eng.device
    imgs = torch.from_numpy(S.candidate_images(128, C, seed=3)).to(dev)      # 128 distinct synthetic scans, reused cyclically
    feats = torch.empty((frames, 360, 128), dtype=torch.float32, device=dev)
    specs = torch.empty((frames, 128, eng.SPEC_W), dtype=torch.float32, device=dev)
    dcs = torch.empty((frames, eng.DELTA_CACHE_ELEMS), dtype=torch.float32, device=dev)

    def run():
        found = 0
        qa.submit(imgs[0:1])
        for i in range(frames):
            if i + 1 < frames:
                qa.submit(imgs[(i + 1) % 128:(i + 1) % 128 + 1])
            fv, sp = qa.take()
            feats[i:i + 1].copy_(fv)
            specs[i:i + 1].copy_(sp)
            eng.delta_cache(feats[i:i + 1], out=dcs[i:i + 1])
            if i == 0:
                continue
            r = eng.heads(feats[:i], feats[i:i + 1], spec_l=specs[:i], spec_r=specs[i:i + 1], dcache_l=dcs[:i])
            found += decode_match(eng.best_match(r["overlap"], r["yaw"], 0.3)) is not None
        return found

    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    found = run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    qa.close()
    eng.close()
    pairs = frames * (frames - 1) // 2
    return {"frames": frames, "pairs": pairs, "seconds": dt, "frames_per_s": frames / dt, "pairs_per_s": pairs / dt,
            "loop_closures_reported": int(found)}


def api_sweep(frames: int = 1101, mode: str = "multiple"):
    from overlapnet_amd.infer import FeatureVolumeCache, Infer
    fx = S.load_fixture_images()
    with tempfile.TemporaryDirectory() as tmp:
        seq = os.path.join(tmp, "07")
        os.makedirs(os.path.join(seq, "depth"))
        os.makedirs(os.path.join(seq, "normal"))
        for s in range(0, frames, 128):
            imgs = S.candidate_images(min(128, frames - s), 4, seed=1234 + s, fixture=fx)
            imgs = np.roll(imgs, (s * 37) % 900, axis=2)
            for k in range(imgs.shape[0]):
                np.save(os.path.join(seq, "depth", "%06d.npy" % (s + k)), imgs[k, :, :, 0])
                np.save(os.path.join(seq, "normal", "%06d.npy" % (s + k)), imgs[k, :, :, 1:4])
        cfg = {"model": dict(S.REFERENCE_MODEL_CFG, inputShape=[64, 900]), "infer_seqs": "07", "data_root_folder": tmp,
               "use_depth": True, "use_normals": True, "use_class_probabilities": False, "use_class_probabilities_pca": False,
               "use_intensity": False, "batch_size": 16, "pretrained_weightsfilename": ""}
        inf = Infer(cfg, weights=S.make_test_weights(4, seed=0))
        for i in range(min(3, frames)):                      # warm-up (scratch allocation, first touch)
            inf.infer_multiple(i, list(range(i)))
        inf.feature_volumes = FeatureVolumeCache(inf.engine)
        torch.cuda.synchronize()
        pairs = 0
        r = None
        t0 = time.perf_counter()
        for i in range(frames):
            refs = list(range(i))
            r = inf.infer_multiple(i, refs) if mode == "multiple" else inf.infer_best_match(i, refs, 0.3)
            pairs += i
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        inf.engine.close()
    return {"frames": frames, "mode": mode, "pairs": pairs, "seconds": dt, "frames_per_s": frames / dt, "pairs_per_s": pairs / dt,
            "last_result": (np.asarray(r[0]).reshape(-1)[:3].tolist() if mode == "multiple" and r is not None else str(r))}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=1101)
    ap.add_argument("--mode", default="multiple", choices=["multiple", "best_match"])
    a = ap.parse_args()
    torch.cuda.set_device(0)
    e = engine_sweep(a.frames)
    p = api_sweep(a.frames, a.mode)
    print(json.dumps({"engine_level": e, "infer_api": p, "api_over_engine": p["frames_per_s"] / e["frames_per_s"]}))
