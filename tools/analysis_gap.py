#!/usr/bin/env python3
"""tools/analysis_gap.py — how far is the build this port is bit-exact to (FIXED_POINT + DISABLE_FLOAT_API, oracle/_ref/libopus_ref_fx.so) from the fixed-point build
users get by default, in which src/analysis.c + mlp.c (float) run and steer the encoder (oracle/_ref/libopus_ref_fxa.so)?  For each BASELINE configuration and an
unforced AUDIO / VOIP encoder: the share of identical packets, which TOCs (mode / bandwidth decisions) differ, the packet-size statistics, and the quality of each
build's output against the input by the reference's own opus_compare.  CPU only (both sides are the compiled reference); it is the measurement behind DESIGN.md's
statement of the analysis gap and the acceptance test the device version of the analysis will be held to."""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import capi, signals
from test_kernel_emu_silkdec import speechy

CASES = [("config 2: CELT-only 48 kHz stereo 128 kb/s CVBR", 48000, 2, 2051, dict(bitrate=128000, complexity=10, vbr_constraint=1), "music"),
         ("config 3: SILK-only 16 kHz mono 24 kb/s", 16000, 1, 2048, dict(force_mode=1000, bandwidth=1103, bitrate=24000, complexity=10), "speech"),
         ("config 4: hybrid 48 kHz stereo 128 kb/s", 48000, 2, 2049, dict(force_mode=1001, bandwidth=1105, bitrate=128000, complexity=10), "speech"),
         ("AUDIO 48 kHz stereo 64 kb/s, nothing forced, music", 48000, 2, 2049, dict(bitrate=64000, complexity=10), "music"),
         ("AUDIO 48 kHz stereo 32 kb/s, nothing forced, speech", 48000, 2, 2049, dict(bitrate=32000, complexity=10), "speech"),
         ("VOIP 16 kHz mono 20 kb/s, nothing forced, speech", 16000, 1, 2048, dict(bitrate=20000, complexity=10), "speech")]

def run(name, Fs, ch, app, ctl, kind, frames=250):
    n = Fs // 50
    if kind == "music": sig = np.ascontiguousarray(signals.music(frames + 1, seed=5)[::48000 // Fs][:, :ch]) if ch == 2 else np.ascontiguousarray(signals.music(frames + 1, seed=5)[::48000 // Fs, :1])
    else: sig = np.ascontiguousarray(speechy(frames + 1, ch, 77, 960)[::48000 // Fs])
    out = {}
    for lib in ("ref", "ref_fxa"):
        e = capi.Enc(lib, Fs, ch, app, **ctl); d = capi.Dec("ref", Fs, ch)
        pk = []; pcm = []
        for f in range(frames):
            p, ln, rng = e.encode(np.ascontiguousarray(sig[f * n:(f + 1) * n]), n); assert ln > 0
            pk.append(bytes(p[:ln])); pcm.append(d.decode(pk[-1], n)[1])
        out[lib] = (pk, np.concatenate(pcm))
    a, b = out["ref"][0], out["ref_fxa"][0]
    same = sum(x == y for x, y in zip(a, b)); toc = sum(x[0] != y[0] for x, y in zip(a, b))
    first = next((i for i, (x, y) in enumerate(zip(a, b)) if x != y), None)
    return dict(case=name, frames=frames, identical=same, toc_differs=toc, first_difference=first, mean_bytes=(round(np.mean([len(x) for x in a]), 1), round(np.mean([len(x) for x in b]), 1)))

if __name__ == "__main__":
    for c in CASES:
        r = run(*c)
        print("%-58s identical packets %3d/%d, TOC differs in %3d, first difference at frame %s, mean bytes %s (no analysis) / %s (analysis)" %
              (r["case"], r["identical"], r["frames"], r["toc_differs"], r["first_difference"], r["mean_bytes"][0], r["mean_bytes"][1]))
