"""BASELINE config 3 emulated (KITTI 07 is not in the tree): a 1101-frame sequence, every frame's scan goes through the leg and
is compared with ALL previous frames (ungated, 605,550 pairs), decision on the device, one record per frame back to the
host -- the streaming use of demo3_lcd.py with the feature / spectrum caches resident in HBM.   python tools/bench_lcd.py
(engine level; tools/bench_infer_api.py runs the same sweep through the `Infer` class as well)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_infer_api import engine_sweep  # noqa: E402

print(json.dumps(engine_sweep(1101)))
