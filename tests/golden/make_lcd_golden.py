#!/usr/bin/env python3
"""Generate tests/golden/lcd_gating.npz by RUNNING the reference's own loop-closure gating: `AnimatedLCD.get_predictions` and
`AnimatedLCD.get_cov_ellipse` of demo/demo3_lcd.py:85-140, imported unmodified.  The module's other imports (Keras behind
`infer`, a Tk matplotlib backend) are not needed by those two methods and are replaced by stubs before the import; the network
itself is replaced by a recorder whose "overlaps" are a fixed function of (current, reference) frame ids -- what is pinned here
is WHICH frames the reference compares and WHICH one it reports, for synthetic trajectories and covariances.

Run in the build container only (needs /root/reference):    python tests/golden/make_lcd_golden.py
"""
import os
import sys
import types

import numpy as np

REF = os.environ.get("OVERLAPNET_REFERENCE", "/root/reference")


def fake_overlap(cur, refs):
    """Deterministic stand-in for the network output: in [0.05, 0.95], peaks where (cur + 3 ref) % 17 is small."""
    refs = np.asarray(refs, np.int64)
    return (0.05 + 0.9 * (((cur + 3 * refs) % 17) / 16.0) ** 2).astype(np.float32)


def fake_yaw(cur, refs):
    return ((np.asarray(refs, np.int64) * 7 + cur) % 360 - 179).astype(np.int64)


class RecorderInfer(object):
    """What get_predictions sees behind `self.infer`: records every call."""

    def __init__(self):
        self.calls = []

    def infer_multiple(self, cur, refs):
        refs = list(np.asarray(refs, np.int64))
        self.calls.append((int(cur), refs))
        if len(refs) == 0:
            return None
        return fake_overlap(cur, refs), fake_yaw(cur, refs)


def import_reference_demo3():
    class Ellipse(object):      # matplotlib.patches.Ellipse stand-in: keeps the constructor arguments
        def __init__(self, xy, width, height, angle=0.0, **kw):
            self.center, self.width, self.height, self.angle = xy, width, height, angle
    mpl = types.ModuleType("matplotlib")
    mpl.use = lambda *a, **k: None
    sub = {name: types.ModuleType("matplotlib." + name) for name in ("pyplot", "lines", "patches", "animation")}
    sub["lines"].Line2D = object
    sub["patches"].Ellipse = Ellipse
    infer_stub = types.ModuleType("infer")      # `from infer import *` (Keras) -- not used by the two methods
    utils_stub = types.ModuleType("utils")      # `from utils import *`: the gating uses numpy through this star import
    utils_stub.np = np
    saved = {k: sys.modules.get(k) for k in ["matplotlib", "infer", "utils"] + ["matplotlib." + n for n in sub]}
    sys.modules.update({"matplotlib": mpl, "infer": infer_stub, "utils": utils_stub})
    for n, m in sub.items():
        sys.modules["matplotlib." + n] = m
        setattr(mpl, n, m)
    sys.path.insert(0, os.path.join(REF, "demo"))
    try:
        import demo3_lcd
    finally:
        sys.path.pop(0)
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return demo3_lcd


def trajectories():
    """name -> (xy (n,2), cov (n,2,2)): an out-and-back drive, a figure of eight, a random walk with growing uncertainty."""
    out = {}
    t = np.arange(420, dtype=np.float64)
    xy = np.stack([np.where(t < 210, t, 419.0 - t) * 0.9, np.where(t < 210, 0.0, 1.5 + 0.01 * t)], axis=1)
    cov = np.tile(np.array([[9.0, 1.5], [1.5, 4.0]]), (len(t), 1, 1)) * (1.0 + t[:, None, None] / 300.0)
    out["out_and_back"] = (xy, cov)
    s = np.linspace(0, 4 * np.pi, 700)
    xy = np.stack([60.0 * np.sin(s), 35.0 * np.sin(2 * s)], axis=1)
    ang = 0.3 * s
    R = np.stack([np.stack([np.cos(ang), -np.sin(ang)], -1), np.stack([np.sin(ang), np.cos(ang)], -1)], -2)
    cov = R @ np.diag([25.0, 2.0]) @ np.transpose(R, (0, 2, 1))
    out["figure_eight"] = (xy, cov)
    rng = np.random.default_rng(7)
    xy = np.cumsum(rng.normal(0, 0.8, size=(500, 2)) + np.array([0.3, 0.0]) * np.sin(np.arange(500) / 40.0)[:, None], axis=0)
    a = rng.normal(size=(500, 2, 2))
    cov = a @ np.transpose(a, (0, 2, 1)) * (2.0 + np.arange(500)[:, None, None] / 50.0)
    out["random_walk"] = (xy, cov)
    return out


def main():
    demo3 = import_reference_demo3()
    cls = demo3.AnimatedLCD
    store = {}
    for name, (xy, cov) in trajectories().items():
        n = len(xy)
        # demo3 keeps the travelled distance in self.traj_length (update(), demo3_lcd.py:147-151: += norm of the pose increment)
        traj_length = np.concatenate([[0.0], np.cumsum(np.linalg.norm(np.diff(xy, axis=0), axis=1))])
        me = types.SimpleNamespace(infer=RecorderInfer(), traj_length=list(traj_length), inactive_time_thres=100,
                                   inactive_dist_thres=50, overlap_thres=0.3)
        ell = np.zeros((n, 3))
        result = np.full(n, -2, np.int64)            # -2: None returned, else the reported reference frame id
        ncall = np.zeros(n, np.int64)
        refs_flat, refs_off = [], [0]
        for idx in range(n):
            e = cls.get_cov_ellipse(me, cov[idx], xy[idx], 3)
            ell[idx] = [e.width, e.height, e.angle]
            before = len(me.infer.calls)
            r = cls.get_predictions(me, idx, xy, e)
            result[idx] = -2 if r is None else int(r)
            ncall[idx] = len(me.infer.calls) - before
            refs = me.infer.calls[-1][1] if ncall[idx] else []
            refs_flat += refs
            refs_off.append(len(refs_flat))
        store[name + "_xy"] = xy
        store[name + "_cov"] = cov
        store[name + "_ellipse"] = ell
        store[name + "_result"] = result
        store[name + "_ncall"] = ncall
        store[name + "_refs"] = np.asarray(refs_flat, np.int64)
        store[name + "_refs_off"] = np.asarray(refs_off, np.int64)
        print(name, "frames", n, "frames with candidates", int(np.sum(np.diff(refs_off) > 0)), "loop closures", int(np.sum(result >= 0)),
              "frames without an infer call", int(np.sum(ncall == 0)))
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lcd_gating.npz")
    np.savez_compressed(out, **store)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
