#!/usr/bin/env python3
"""tools/gen_bit_vectors.py — config 1 stand-in vectors: the RFC 8251 test vectors are not on this box (no network), so this script produces `.bit` files in
opus_demo's framing (src/opus_demo.c:1102-1112: per packet a 4-byte big-endian length, the 4-byte big-endian encoder final range, the payload) with the COMPILED
REFERENCE ENCODER over the mode matrix of tests/test_opus_encode.c:330-512 (modes x bandwidths x frame sizes, mono / stereo, VBR / CBR, in-band FEC, DTX: 13 hand-picked files, and -- round 6, `mx_*` -- the 3 x 13 rows of that test's own tables).  Runs where
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
# the Encode+Decode matrix of tests/test_opus_encode.c:420-512 itself: rc = 0 VBR + in-band FEC, 1 constrained VBR, 2 hard CBR; the 13 (mode, rate, frame) rows of its tables;
# bandwidth, complexity, DTX and loss percentage -- drawn at random there -- walk through their ranges deterministically here; force_channels by rate as there
REF_MODES = [0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 2, 2]
REF_RATES = [6000, 12000, 48000, 16000, 32000, 48000, 64000, 512000, 13000, 24000, 48000, 64000, 96000]
REF_FRAME = [960 * 2, 960, 480, 960, 960, 960, 480, 960 * 3, 960 * 3, 960, 480, 240, 120]
for rc in range(3):
    for j in range(13):
        mode = REF_MODES[j]
        bw = (1101 + (j + rc) % 3) if mode == 0 else (1104 + (j + rc) % 2) if mode == 1 else (1101 + (2 * j + rc) % 5)
        if bw == 1102 and mode == 2: bw = 1101                                   # (CELT has no mediumband)
        MATRIX.append(("mx_rc%d_%s_%d_%d" % (rc, ("silk", "hybrid", "celt")[mode], REF_RATES[j], REF_FRAME[j]), 2049, 2, REF_FRAME[j],
                       dict(vbr=int(rc < 2), vbr_constraint=int(rc == 1), inband_fec=int(rc == 0), force_mode=1000 + mode, dtx=(j + rc) & 1, bitrate=REF_RATES[j] + (7919 * (j + 3 * rc)) % REF_RATES[j],
                            force_channels=2 if REF_RATES[j] >= 64000 else 1, complexity=(3 * j + 5 * rc) % 11, packet_loss=(5 * j + rc) % 15, bandwidth=bw)))
SWITCH = {"auto_switching_stereo": {10: dict(bitrate=96000), 20: dict(bitrate=16000, force_mode=1000), 30: dict(force_mode=1002), 40: dict(force_mode=1001, bandwidth=1105, bitrate=64000)}}

def main(outdir=os.path.join(ROOT, "tests/golden/bitstreams"), seconds=1.0):
    os.makedirs(outdir, exist_ok=True)
    for name, app, ch, fr, ctl in MATRIX:
        n = max(8, int((0.3 if name.startswith("mx_") else seconds) * 48000 / fr))
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
