#!/usr/bin/env python3
"""Run another tool with development knobs set first (libhsefr_dev.so):   KNOBS=w4_off=2,c11_bres=0 python tools/run_knob.py tools/bench_configs.py resnet50"""
import os, sys, runpy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HSEFR_LIB", "libhsefr_dev.so")
from hse_facerec_tf_amd import _lib
for kv in os.environ.get("KNOBS", "").split(","):
    if kv:
        k, v = kv.split("=")
        _lib.check(_lib.lib().hsefr_debug_set(k.encode(), int(v)))
sys.argv = sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")
