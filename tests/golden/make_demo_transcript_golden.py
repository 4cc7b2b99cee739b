#!/usr/bin/env python3
"""tests/golden/demo_transcript.json: what the reference's OWN demo scripts do with the `Infer` object -- demo/demo2_infer.py
(`demo_infer`, :15-47, driven like its __main__, :50-69) and demo/demo3_lcd.py (`AnimatedLCD.__init__/setup_plot/update/
get_predictions`, :23-176), both imported UNMODIFIED and run here with `sys.modules['infer']` replaced by a recorder (Keras is not
installable) and matplotlib / the Tk backend replaced by inert stand-ins.  Recorded: the constructor's config, every method call
with argument values AND types, every attribute the scripts read, and what the scripts did with the returned values.
tests/test_gpu_api.py::test_reference_demos_replayed_on_the_drop_in replays the transcript against `overlapnet_amd.infer.Infer`
on the GPU: the drop-in must accept exactly these calls and return objects the scripts' own code paths work on.

Build container only (needs /root/reference):    python tests/golden/make_demo_transcript_golden.py
"""
import copy
import json
import os
import sys
import types

import numpy as np
import yaml as real_yaml

REF = os.environ.get("OVERLAPNET_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
EVENTS = []


def typename(v):
    if isinstance(v, np.ndarray):
        return "ndarray[%s]%s" % (v.dtype, list(v.shape))
    if isinstance(v, (list, tuple)):
        return "%s[%d]" % (type(v).__name__, len(v))
    return type(v).__name__


def plain(v):
    if isinstance(v, np.ndarray):
        return v.tolist()
    if isinstance(v, (np.integer,)):
        return int(v)
    if isinstance(v, (list, tuple)):
        return [plain(x) for x in v]
    return v


def fake_overlap(cur, refs):
    refs = np.asarray(refs, np.int64)
    return (0.05 + 0.9 * (((cur + 3 * refs) % 17) / 16.0) ** 2).astype(np.float32)


class RecorderInfer(object):
    """Stands where `from infer import *` puts the reference's class (src/two_heads/infer.py:22)."""

    def __init__(self, config):
        EVENTS.append({"event": "Infer", "config": copy.deepcopy(config)})
        object.__setattr__(self, "_attrs", {"datasetpath": config["data_root_folder"], "seq": config["infer_seqs"], "filenames": None})

    def __getattr__(self, name):
        attrs = object.__getattribute__(self, "_attrs")
        if name in attrs:
            EVENTS.append({"event": "getattr", "name": name})
            return attrs[name]
        raise AttributeError(name)

    def infer_one(self, filepath1, filepath2):
        EVENTS.append({"event": "infer_one", "args": [filepath1, filepath2], "types": [typename(filepath1), typename(filepath2)]})
        n1 = os.path.basename(filepath1).replace(".bin", "")
        n2 = os.path.basename(filepath2).replace(".bin", "")
        object.__getattribute__(self, "_attrs")["filenames"] = np.array([n2, n1])      # infer.py:140
        return np.array([0.8919191], np.float32), np.array([0])                       # shapes of infer.py:157-160

    def infer_multiple(self, cur, refs):
        EVENTS.append({"event": "infer_multiple", "cur": plain(cur), "cur_type": typename(cur), "refs": plain(refs), "refs_type": typename(refs)})
        if len(refs) == 0:
            return None
        ov = fake_overlap(int(cur), refs)
        yaw = ((np.asarray(refs, np.int64) * 7 + int(cur)) % 360 - 179).astype(np.int64)
        return ov.reshape(-1, 1).squeeze(), yaw                                       # infer.py:197-200 (0-d when one reference)


class Anything(object):
    """Inert stand-in for every matplotlib object: any attribute is callable and returns another stand-in; iterable of one
    (`line, = ax.plot(...)`)."""
    def __init__(self, *a, **k):
        for key, v in k.items():
            object.__setattr__(self, key, v)
        if a and not k:
            pass

    def __getattr__(self, name):
        return Anything()

    def __call__(self, *a, **k):
        return Anything()

    def __iter__(self):
        return iter((Anything(),))

    def __getitem__(self, i):
        return Anything()


class Ellipse(object):
    def __init__(self, xy=(0, 0), width=0, height=0, angle=0.0, **kw):
        self.center, self.width, self.height, self.angle = xy, width, height, angle


def install_stubs():
    mpl = types.ModuleType("matplotlib")
    mpl.use = lambda *a, **k: EVENTS.append({"event": "matplotlib.use", "backend": a[0] if a else None})
    names = ("pyplot", "lines", "patches", "animation", "gridspec")
    sub = {n: types.ModuleType("matplotlib." + n) for n in names}
    plt = sub["pyplot"]
    for fn in ("figure", "subplot", "suptitle", "show", "subplots"):
        setattr(plt, fn, Anything())
    plt.subplots = lambda *a, **k: (Anything(), Anything())
    sub["lines"].Line2D = Anything
    sub["patches"].Ellipse = Ellipse
    sub["animation"].FuncAnimation = Anything
    sub["gridspec"].GridSpec = Anything
    y = types.ModuleType("yaml")                       # the scripts call yaml.load(f) without a Loader (PyYAML < 6)
    y.load = lambda f, *a, **k: real_yaml.safe_load(f)
    inf = types.ModuleType("infer")
    inf.Infer = RecorderInfer
    sys.modules.update({"matplotlib": mpl, "yaml": y, "infer": inf})
    for n, m in sub.items():
        sys.modules["matplotlib." + n] = m
        setattr(mpl, n, m)


def main():
    install_stubs()
    os.chdir(REF)                                      # the demos use paths relative to the repository root (config/demo.yml)
    sys.path.insert(0, os.path.join(REF, "demo"))
    out = {"reference": "PRBonn/OverlapNet demo/demo2_infer.py + demo/demo3_lcd.py, imported unmodified"}

    # ---- demo2, as its __main__ drives it (:50-69) ------------------------------------------------------------------------
    import demo2_infer
    config = real_yaml.safe_load(open("config/demo.yml"))
    network_config = real_yaml.safe_load(open(config["Demo2"]["network_config"]))
    network_config["infer_seqs"] = config["Demo2"]["infer_seqs"]
    EVENTS.clear()
    demo2_infer.demo_infer(network_config, config["Demo2"]["scan2_path"], config["Demo2"]["scan1_path"])
    out["demo2"] = list(EVENTS)
    out["demo2_depth_files_loaded"] = ["data/preprocess_data_demo/depth/000000.npy", "data/preprocess_data_demo/depth/000001.npy"]

    # ---- demo3: the class as __main__ builds it (:222), stepped through its own update() ------------------------------------
    import demo3_lcd
    n = 260
    t = np.arange(n, dtype=np.float64)
    poses = np.tile(np.eye(4), (n, 1, 1))
    poses[:, 0, 3] = np.where(t < 130, t, 259.0 - t) * 0.9          # out and back: the way back passes the way out
    poses[:, 1, 3] = np.where(t < 130, 0.0, 1.0)
    covs = np.zeros((n, 36))
    covs[:, 0], covs[:, 7], covs[:, 1], covs[:, 6] = 9.0, 4.0, 1.5, 1.5
    demo3_lcd.covs = covs                                            # data_stream reads the module-level name (:146)
    EVENTS.clear()
    lcd = demo3_lcd.AnimatedLCD(config["Demo3"]["network_config"], poses, covs)
    lcd.setup_plot()
    closures = []
    lcd.loop_closure = types.SimpleNamespace(set_offsets=lambda xy: closures.append(plain(np.asarray(xy))))
    for i in range(n - 1):
        lcd.update(i)
    out["demo3"] = list(EVENTS)
    out["demo3_frames"] = n - 1
    out["demo3_loop_closures_with_the_recorders_scores"] = len(closures)
    calls = [e for e in EVENTS if e["event"] == "infer_multiple"]
    print("demo2 events:", [e["event"] for e in out["demo2"]])
    print("demo3: %d infer_multiple calls, %d with references, argument types %s, %d loop closures" % (
        len(calls), sum(1 for c in calls if c["refs"]), sorted({(c["cur_type"], c["refs_type"].split("[")[0]) for c in calls}), len(closures)))
    json.dump(out, open(os.path.join(HERE, "demo_transcript.json"), "w"))
    print("wrote", os.path.join(HERE, "demo_transcript.json"))


if __name__ == "__main__":
    main()
