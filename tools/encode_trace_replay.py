#!/usr/bin/env python3
"""tools/encode_trace_replay.py <context log> <pcm dump> <first call> <last call> [which=emu] — replay one encoder's life out of a tools/encode_trace_shim.c log
(its creation, every control, every opus_encode call with the dumped input) on the compiled reference (float API on) and on this library (`emu` or `gpu`), and
report the first calls whose packets differ, with the settings in force.  The dump comes from a run of the reference-linked program with
OPUS_TRACE_PCM_FROM / _TO / OPUS_TRACE_PCM set."""
import sys, os, ctypes, struct, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("OPUS_AMD_FLOAT_ANALYSIS", "1")
import capi
ctx, dump, lo, hi = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
which = sys.argv[5] if len(sys.argv) > 5 else "emu"
pcm = {}
with open(dump, "rb") as f:
    while True:
        h = f.read(16)
        if len(h) < 16: break
        n, fs, ch, maxb = struct.unpack("<4i", h)
        pcm[n] = (fs, ch, maxb, np.frombuffer(f.read(2 * fs * ch), np.int16).copy())
events = []; create = None; last = -1
for l in open(ctx):
    t = l.split()
    if l.startswith("# create"): cur = (int(t[3][3:]), int(t[4][3:]), int(t[5][4:])); cur_at = last + 1
    elif l.startswith("# ctl"): events.append(("ctl", last + 1, int(t[3]), int(t[4])))
    elif not l.startswith("#"):
        last = int(t[0])
        if last == lo: create = cur; assert cur_at == lo, (cur_at, lo)
        if lo <= last <= hi: events.append(("enc", last))
events = [e for e in events if lo <= e[1] <= hi + 1]
Fs, ch, app = create
print("encoder", create, "calls", lo, hi)
R = capi.Enc("ref_fxa", Fs, ch, app); E = capi.Enc(which, Fs, ch, app)
names = {v: k for k, v in capi.REQ.items()}
cur = {}; nbad = 0
for e in events:
    if e[0] == "ctl":
        for X in (R, E):
            X.L.opus_encoder_ctl.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]; r = X.L.opus_encoder_ctl(X.st, e[2], e[3])
        cur[names.get(e[2], e[2])] = e[3]
    else:
        fs, c, maxb, x = pcm[e[1]]
        a = R.encode(x, fs, maxb); b = E.encode(x, fs, maxb)
        if a != b:
            nbad += 1
            if nbad <= 5: print("call", e[1], "fs", fs, "max", maxb, "ref", a[1], "%02x" % (a[0][0] if a[0] else 0), "got", b[1], "%02x" % (b[0][0] if b[0] else 0), cur)
print("differing calls:", nbad)
