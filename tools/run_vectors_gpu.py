#!/usr/bin/env python3
"""tools/run_vectors_gpu.py — BASELINE config 1 on the MI355X: the equivalent of tests/run_vectors.sh for this library.  Reads `.bit` files in opus_demo's framing
(src/opus_demo.c:963-976: 4-byte big-endian length, 4-byte big-endian encoder final range, payload), decodes ALL files together as the streams of ONE batch
(opusgpu_decode_batch: one wavefront per file per step), and checks per packet
  * OPUS_GET_FINAL_RANGE against the range stored in the file (src/opus_demo.c:1217-1226),
  * the PCM byte for byte against the compiled reference decoder (fixed-point build, oracle/_ref/libopus_ref_fx.so) -- the "bit-exact in fixed-point mode" gate,
  * and, when <name>.dec files and an opus_compare binary are supplied (RFC 8251 vectors), opus_compare's verdict on the written PCM.
usage: run_vectors_gpu.py [directory with *.bit] [--rate 48000] [--channels 2] [--lib gpu|emu]"""
import argparse, ctypes, glob, os, struct, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

def read_bit(path):
    pk = []; data = open(path, "rb").read(); p = 0
    while p + 8 <= len(data):
        ln, rng = struct.unpack(">II", data[p:p + 8]); p += 8
        if ln > 1500 or p + ln > len(data): raise ValueError("%s: invalid payload length %d" % (path, ln))
        pk.append((data[p:p + ln], rng)); p += ln
    return pk

def run(files, Fs=48000, channels=2, which="gpu", outdir=None, compare=None):
    import capi
    L = capi.load(which)
    vp, i32 = ctypes.c_void_p, ctypes.c_int32
    L.opusgpu_dec_batch_create.restype = vp; L.opusgpu_dec_batch_create.argtypes = [i32, i32, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    L.opusgpu_decode_batch.argtypes = [vp, vp, i32, vp, vp, ctypes.c_int, vp, vp]; L.opusgpu_dec_batch_destroy.argtypes = [vp]
    streams = [read_bit(f) for f in files]
    S = len(streams); err = ctypes.c_int()
    b = L.opusgpu_dec_batch_create(S, Fs, channels, 0, ctypes.byref(err)); assert b and err.value == 0, err.value
    refs = [capi.Dec("ref", Fs, channels) for _ in range(S)]
    maxfr = Fs // 25 * 3; stride = 1536
    pcm_out = [[] for _ in range(S)]
    bad = 0; npk = 0
    for t in range(max(len(s) for s in streams)):
        pk = np.zeros((S, stride), np.uint8); lens = np.zeros(S, np.int32)
        for s in range(S):
            if t < len(streams[s]): d = streams[s][t][0]; pk[s, :len(d)] = np.frombuffer(d, np.uint8); lens[s] = len(d)
        pcm = np.zeros((S, maxfr * channels), np.int16); ns = np.zeros(S, np.int32); rng = np.zeros(S, np.uint32)
        r = L.opusgpu_decode_batch(b, pk.ctypes.data, stride, lens.ctypes.data, pcm.ctypes.data, maxfr, ns.ctypes.data, rng.ctypes.data); assert r == 0, r
        for s in range(S):
            if t >= len(streams[s]): continue
            d, enc_rng = streams[s][t]; npk += 1
            n, ref_pcm, ref_rng = refs[s].decode(d, maxfr)
            ok = int(ns[s]) == n and int(rng[s]) == ref_rng and (len(d) <= 1 or ref_rng == enc_rng or enc_rng == 0) and np.array_equal(pcm[s, :n * channels].reshape(n, channels), ref_pcm)
            if not ok:
                bad += 1
                print("MISMATCH %s packet %d: samples %d/%d range %08x/%08x/%08x" % (os.path.basename(files[s]), t, int(ns[s]), n, int(rng[s]), ref_rng, enc_rng))
            pcm_out[s].append(pcm[s, :max(int(ns[s]), 0) * channels].copy())
    L.opusgpu_dec_batch_destroy(b)
    verdicts = {}
    if outdir:
        os.makedirs(outdir, exist_ok=True)
        for s, f in enumerate(files):
            o = os.path.join(outdir, os.path.basename(f)[:-4] + ".pcm"); np.concatenate(pcm_out[s]).tofile(o)
            dec = f[:-4] + ".dec"
            if compare and os.path.exists(dec):
                p = subprocess.run([compare] + (["-s"] if channels == 2 else []) + ["-r", str(Fs), dec, o], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
                verdicts[os.path.basename(f)] = p.returncode == 0
    return npk, bad, verdicts

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("dir", nargs="?", default=os.path.join(ROOT, "tests/golden/bitstreams"))
    ap.add_argument("--rate", type=int, default=48000); ap.add_argument("--channels", type=int, default=2); ap.add_argument("--lib", default="gpu")
    ap.add_argument("--out", default=None); ap.add_argument("--opus-compare", default=None)
    a = ap.parse_args()
    files = sorted(glob.glob(os.path.join(a.dir, "*.bit")))
    npk, bad, verdicts = run(files, a.rate, a.channels, a.lib, a.out, a.opus_compare)
    print("%d files, %d packets, %d mismatches%s" % (len(files), npk, bad, "" if not verdicts else ", opus_compare: %d/%d pass" % (sum(verdicts.values()), len(verdicts))))
    sys.exit(1 if bad or not all(verdicts.values()) else 0)
