#!/usr/bin/env python3
"""tools/gpu_micro_check.py — a parity check of the SILK-capable encoder's kernel pipeline that fits in a few seconds of GPU time: config 3 (80 streams x 4 frames) and
config 4 (64 streams x 3 frames), every packet, length and final range against the compiled reference (the check function of tests/test_gpu_silkenc.py; launches of 64
streams and more take the front / quantiser / back kernels).  numpy + ctypes only."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("OPUS_AMD_FLOAT_ANALYSIS", "0")          # the checker of that function is the reference built without the float API (tests/conftest.py does the same)
t0 = time.time()
from test_gpu_silkenc import check
check(80, 4, bitrate=24000, complexity=10, force_mode=1000, bandwidth=1103)
print("config 3 ok  %.1f s" % (time.time() - t0), flush=True)
check(64, 3, Fs=48000, ch=2, app=2049, bitrate=128000, complexity=10, force_mode=1001, bandwidth=1105)
print("config 4 ok  %.1f s" % (time.time() - t0), flush=True)
import opus_amd
print("MICRO OK", opus_amd.lib().opusgpu_build_info().decode() if hasattr(opus_amd.lib(), "opusgpu_build_info") else "")
