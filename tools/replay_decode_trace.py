#!/usr/bin/env python3
"""tools/replay_decode_trace.py — replay the opus_decode calls recorded by tools/decode_trace_shim.c (LD_PRELOAD under any program linked to this library, e.g. the
reference's test_opus_encode on the emulated library) through this decoder (emulated library) and the compiled reference decoder side by side: sample count, PCM and
final range per call.  usage: replay_decode_trace.py trace.bin [decoder pointer ...]
   gcc -O2 -shared -fPIC tools/decode_trace_shim.c -o /tmp/dec_trace.so -ldl
   OPUS_TRACE_DUMP=/tmp/trace.bin LD_PRELOAD=/tmp/dec_trace.so oracle/_ref/reftests/emu/test_opus_encode"""
import sys, struct, ctypes, numpy as np
import os as _os
_R = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
sys.path.insert(0, _os.path.join(_R, 'tests')); sys.path.insert(0, _R)
import capi, os, hostemu
if os.environ.get("OLD"): hostemu.build_emu_lib = lambda *a, **k: "/tmp/old_src/libopus_amd_emu.so"
data = open(sys.argv[1], 'rb').read()
recs = []; p = 0
while p + 32 <= len(data):
    st, = struct.unpack_from('<Q', data, p); kind, a, b, c, d, n = struct.unpack_from('<6i', data, p + 8); p += 32
    if p + n > len(data): break
    recs.append((chr(kind), st, a, b, c, d, data[p:p + n])); p += n
print(len(recs), "records")
cfg = {}
for r in recs:
    if r[0] == 'C': cfg[r[1]] = (r[2], r[3])
calls = {}
for i, r in enumerate(recs):
    if r[0] in 'DR': calls.setdefault(r[1], []).append((i, r))
# decoders that were not made by opus_decoder_create (memcpy copies): MAP="skip,48000:1,48000:2,..." assigns (Fs, channels) -- or skips them -- in order of first use
_map = [m for m in os.environ.get("MAP", "").split(",") if m]
for st in calls:
    if st not in cfg and _map:
        m = _map.pop(0)
        cfg[st] = None if m == "skip" else tuple(int(v) for v in m.split(":"))
for st, cl in calls.items(): print(hex(st), cfg.get(st, (48000, 2, 'copy')), len(cl), "calls")
which = sys.argv[2:] or [hex(k) for k in calls]
for st, cl in calls.items():
    if hex(st) not in which or (st in cfg and cfg[st] is None): continue
    Fs, ch = cfg.get(st, (48000, 2))[:2]
    a = capi.Dec("emu", Fs, ch); b = capi.Dec("ref", Fs, ch)
    bad = 0
    for k, (i, r) in enumerate(cl):
        if r[0] == 'R':
            for d in (a, b):
                d.L.opus_decoder_ctl.argtypes = [ctypes.c_void_p, ctypes.c_int]; d.L.opus_decoder_ctl(d.st, 4028)
            continue
        _, _, ln, fs, fec, ret, pkt = r
        if fs <= 0 or ln < 0:                                                    # argument checks of the test programs: compare the return codes only
            dummy = np.zeros(max(16, fs * ch + 16), np.int16)            # (NULL data with a negative length is a loss: the call may well produce fs samples)
            rx = a.L.opus_decode(a.st, pkt if ln > 0 else None, ln, dummy.ctypes.data, fs, fec); ry = b.L.opus_decode(b.st, pkt if ln > 0 else None, ln, dummy.ctypes.data, fs, fec)
            if rx != ry: bad += 1; print("decoder", hex(st), (Fs, ch), "call", k, "argument check: ret emu/ref", rx, ry)
            continue
        x = a.decode(pkt if ln > 0 else b"", fs, fec); y = b.decode(pkt if ln > 0 else b"", fs, fec)
        same = x[0] == y[0] and x[2] == y[2] and np.array_equal(x[1], y[1])
        if not same:
            bad += 1
            if bad <= 3: print("decoder", hex(st), (Fs, ch), "call", k, "record", i, "len", ln, "fs", fs, "fec", fec, "ret emu/ref", x[0], y[0], "rng", hex(x[2]), hex(y[2]), "pcm equal", np.array_equal(x[1], y[1]), "toc", pkt[:1].hex(), "orig ret", ret)
            if bad == 1: open('/tmp/bad_packet_%x_%d.bin' % (st, k), 'wb').write(pkt)
    print("decoder", hex(st), (Fs, ch), "calls", len(cl), "mismatching calls", bad)
sys.stdout.flush(); os._exit(0)          # (skip interpreter teardown: the ctypes decoder handles are plain integers)
