#!/usr/bin/env python3
"""tools/classic_threads_bench.py — what concurrent callers of the classic API get (opus_encode / opus_decode on caller-owned states from T host threads): calls per
second and calls per launch for T = 1 .. 256, BASELINE config 2 settings.  The calls of threads that wait together share a launch (opus_amd/csrc/opus_call_combiner.h);
OPUS_AMD_CLASSIC_BATCH=1 gives the one-launch-per-call behaviour of rounds 1-2 for comparison.  Python threads: ctypes releases the GIL for the duration of a call."""
import ctypes, json, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("OPUS_AMD_FLOAT_ANALYSIS", "1")
import numpy as np, capi, signals

def stats():
    L = capi.load("gpu"); out = (ctypes.c_longlong * 4)()
    L.opusgpu_classic_call_stats.argtypes = [ctypes.c_void_p]; L.opusgpu_classic_call_stats.restype = None
    L.opusgpu_classic_call_stats(out); return list(out)

def run(T, nf, decode):
    xs = [signals.music(nf, seed=k) for k in range(min(T, 16))]
    encs = [capi.Enc("gpu", 48000, 2, 2051, bitrate=128000, complexity=10) for _ in range(T)]
    pk = None
    if decode:
        e = capi.Enc("gpu", 48000, 2, 2051, bitrate=128000, complexity=10)
        pk = [e.encode(xs[0][i * 960:(i + 1) * 960], 960)[0] for i in range(nf)]
        decs = [capi.Dec("gpu", 48000, 2) for _ in range(T)]
    def work(k):
        x = xs[k % len(xs)]
        for i in range(nf):
            if decode: decs[k].decode(pk[i], 960)
            else: encs[k].encode(x[i * 960:(i + 1) * 960], 960)
    work(0) if T == 1 else None                                                 # warm
    s0 = stats(); t0 = time.time()
    th = [threading.Thread(target=work, args=(k,)) for k in range(T)]
    for t in th: t.start()
    for t in th: t.join()
    dt = time.time() - t0; s1 = stats()
    c, l = (s1[2] - s0[2], s1[3] - s0[3]) if decode else (s1[0] - s0[0], s1[1] - s0[1])
    return {"threads": T, "op": "opus_decode" if decode else "opus_encode", "calls_per_s": round(c / dt, 1), "calls_per_launch": round(c / max(l, 1), 2), "ms_per_call_seen_by_a_thread": round(1e3 * dt / nf, 2)}

if __name__ == "__main__":
    nf = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    run(1, 3, False); run(1, 3, True)
    for decode in (False, True):
        for T in (1, 4, 16, 64, 256):
            print(json.dumps(run(T, nf, decode)), flush=True)
