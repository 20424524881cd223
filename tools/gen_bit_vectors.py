#!/usr/bin/env python3
"""tools/gen_bit_vectors.py — config 1 stand-in vectors: the RFC 8251 test vectors are not on this box (no network), so this script produces `.bit` files in
opus_demo's framing (src/opus_demo.c:1102-1112: per packet a 4-byte big-endian length, the 4-byte big-endian encoder final range, the payload) with the COMPILED
REFERENCE ENCODER over the mode matrix of tests/test_opus_encode.c:330-512 (modes x bandwidths x frame sizes, mono / stereo, VBR / CBR, in-band FEC, DTX).  Runs where
oracle/_ref/libopus_ref_fx.so exists; the small files it writes to tests/golden/bitstreams/ are committed (they travel to the GPU box) together with this generator.
When real vectors are supplied, tools/run_vectors_gpu.py takes their directory instead."""
import os, struct, sys, zlib, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import capi
from test_kernel_emu_silkdec import speechy
import signals

MATRIX = [  # name, application, channels, frame (samples @48k), ctls
    ("silk_nb_20ms_mono", 2048, 1, 960, dict(force_mode=1000, bandwidth=1101, bitrate=12000)),
    ("silk_mb_40ms_mono", 2048, 1, 1920, dict(force_mode=1000, bandwidth=1102, bitrate=16000)),
    ("silk_wb_60ms_stereo", 2048, 2, 2880, dict(force_mode=1000, bandwidth=1103, bitrate=36000)),
    ("silk_wb_10ms_fec", 2048, 1, 480, dict(force_mode=1000, bandwidth=1103, bitrate=24000, inband_fec=1, packet_loss=15)),
    ("silk_wb_20ms_dtx", 2048, 1, 960, dict(force_mode=1000, bandwidth=1103, bitrate=20000, dtx=1)),
    ("hybrid_swb_20ms_mono", 2049, 1, 960, dict(force_mode=1001, bandwidth=1104, bitrate=32000)),
    ("hybrid_fb_10ms_stereo", 2049, 2, 480, dict(force_mode=1001, bandwidth=1105, bitrate=64000)),
    ("hybrid_fb_20ms_cbr", 2049, 2, 960, dict(force_mode=1001, bandwidth=1105, bitrate=48000, vbr=0)),
    ("celt_fb_20ms_stereo", 2051, 2, 960, dict(bitrate=128000)),
    ("celt_wb_2p5ms_mono", 2051, 1, 120, dict(bitrate=64000, bandwidth=1103)),
    ("celt_nb_5ms_mono", 2051, 1, 240, dict(bitrate=32000, bandwidth=1101)),
    ("celt_fb_60ms_stereo", 2051, 2, 2880, dict(bitrate=96000)),
    ("auto_switching_stereo", 2049, 2, 960, dict(bitrate=24000)),
]
SWITCH = {"auto_switching_stereo": {10: dict(bitrate=96000), 20: dict(bitrate=16000, force_mode=1000), 30: dict(force_mode=1002), 40: dict(force_mode=1001, bandwidth=1105, bitrate=64000)}}

def main(outdir=os.path.join(ROOT, "tests/golden/bitstreams"), seconds=1.0):
    os.makedirs(outdir, exist_ok=True)
    for name, app, ch, fr, ctl in MATRIX:
        n = max(8, int(seconds * 48000 / fr))
        sig = speechy(n * fr // 960 + 2, ch, zlib.crc32(name.encode()) % 97, 960) if app != 2051 else signals.music(n * fr // 960 + 2, 960, ch, zlib.crc32(name.encode()) % 97)
        e = capi.Enc("ref", 48000, ch, app, **ctl)
        with open(os.path.join(outdir, name + ".bit"), "wb") as f:
            for i in range(n):
                for k, v in SWITCH.get(name, {}).get(i, {}).items(): e.set(k, v)
                pkt, ln, rng = e.encode(sig[i * fr:(i + 1) * fr], fr, 1500)
                assert ln > 0
                f.write(struct.pack(">II", ln, rng)); f.write(pkt)
        print(name, n, "packets")
if __name__ == "__main__":
    main(*sys.argv[1:2])
