#!/usr/bin/env python3
"""tools/silk_enc_bench.py — BASELINE config 3 on one MI355X: SILK-only encode, VOIP 16 kHz mono 20 ms, complexity 10, 24 kb/s VBR, 65,536 streams,
state carried in HBM across steps, inputs resident in HBM, HIP events around K launches.  Prints one JSON line with the roofline figures and the
compiled reference (float build, one host core) timed on the same signal.  (bench.py stays on the headline metric, config 2.)"""
import argparse, ctypes, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

def speech(fs, n, seed):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / fs
    f0 = 120 + 30 * np.sin(2 * np.pi * 0.7 * t + seed) + (seed % 7) * 9
    ph = 2 * np.pi * np.cumsum(f0) / fs
    s = sum(np.sin(k * ph) / k for k in range(1, 25)) * (np.sin(2 * np.pi * 1.5 * t + seed) > -0.3) * 6000 + rng.normal(0, 120, n)
    return np.clip(s, -32768, 32767).astype(np.int16)

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=65536); ap.add_argument("--steps", type=int, default=20); ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--complexity", type=int, default=10); ap.add_argument("--bitrate", type=int, default=24000); ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--fs", type=int, default=16000); ap.add_argument("--channels", type=int, default=1)
    ap.add_argument("--mode", choices=["silk", "hybrid"], default="silk", help="hybrid = BASELINE config 4: AUDIO 48 kHz stereo, forced hybrid fullband, 128 kb/s VBR")
    a = ap.parse_args()
    if a.mode == "hybrid": a.fs, a.channels = 48000, 2; a.bitrate = 128000 if a.bitrate == 24000 else a.bitrate
    import torch, opus_amd
    from reflib import ref_fl, ref_fx
    dev = torch.device("cuda:0")
    S, K, W, Fs, ch = a.streams, a.steps, a.warmup, a.fs, a.channels
    n = Fs // 50
    U = 256
    app = 2049 if a.mode == "hybrid" else 2048
    ctls = ((11002, 1001), (4008, 1105), (4002, a.bitrate), (4010, a.complexity)) if a.mode == "hybrid" else ((11002, 1000), (4008, 1103), (4002, a.bitrate), (4010, a.complexity))
    if ch == 1: base = np.stack([speech(Fs, (W + K) * n, 100 + u) for u in range(U)])                  # [U, steps*n*ch]
    else: base = np.stack([np.stack([speech(Fs, (W + K) * n, 100 + u), speech(Fs, (W + K) * n, 900 + u)], 1).reshape(-1) for u in range(U)])
    d_base = torch.from_numpy(base).to(dev).view(U, W + K, n * ch)
    idx = torch.arange(S, device=dev) % U
    d_pcm = d_base[idx].permute(1, 0, 2).contiguous()                                                # [step][S][n*ch]
    b = opus_amd.EncoderBatch(S, channels=ch, application=app, Fs=Fs)
    for req, v in ctls: b.ctl(req, v)
    d_out = torch.zeros((S, 1280), dtype=torch.uint8, device=dev); d_len = torch.zeros(S, dtype=torch.int32, device=dev); d_rng = torch.zeros(S, dtype=torch.int32, device=dev)
    step_elems = S * n * ch
    b.time_encode_dev(d_pcm.data_ptr(), n, d_out.data_ptr(), 1280, d_len.data_ptr(), d_rng.data_ptr(), W)
    torch.cuda.synchronize()
    ms = b.time_encode_dev(d_pcm.data_ptr() + 2 * W * step_elems, n, d_out.data_ptr(), 1280, d_len.data_ptr(), d_rng.data_ptr(), K) / K
    lens = d_len.cpu().numpy(); rng = d_rng.cpu().numpy().view(np.uint32); out = d_out.cpu().numpy()
    ok = bool((lens > 0).all())
    L = opus_amd.lib()
    silk_state = 14732 if ch == 2 else 14732 - 7344                                                      # OaSilkEnc (mono batches move one channel)
    state_bytes = 4 * 24 + 4 * 44 + silk_state + (10032 + 1920 if a.mode == "hybrid" else 0)            # cfg + scalars + SILK (+ CELT state and delay line), in and out
    bytes_per = n * ch * 2 * (3 if True else 1) + float(lens.mean()) + 8 + 2 * state_bytes               # PCM in + high-passed copy out and in again + packet + state in/out
    res = {"metric": "encoded frames/s (%s, %d kHz %s, 20 ms, complexity %d)" % ("hybrid" if a.mode == "hybrid" else "SILK-only", Fs // 1000, "mono" if ch == 1 else "stereo", a.complexity), "kernel": "oa_sh_encode_kernel",
           "streams": S, "steps": K, "warmup": W, "ms_per_step": ms, "value": S / (ms * 1e-3), "unit": "frames/s", "all_frames_ok": ok, "mean_packet_bytes": float(lens.mean()),
           "lds_bytes_per_wave": L.opusgpu_sh_kernel_lds_bytes(),
           "roofline": {"bound": "hbm", "achieved": S * bytes_per / (ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": S * bytes_per / (ms * 1e-3) / 1e9 / 8000.0, "traffic": None,
                        "bytes_per_frame": bytes_per}}
    # parity spot check + CPU baseline on the same signal: the compiled reference, stream 0
    for name, R in (("fx", ref_fx()), ("fl", ref_fl())):
        if R is None: continue
        R.opus_encoder_create.restype = ctypes.c_void_p
        err = ctypes.c_int(0)
        enc = ctypes.c_void_p(R.opus_encoder_create(Fs, ch, app, ctypes.byref(err)))
        for req, v in ctls: R.opus_encoder_ctl(enc, req, ctypes.c_int(v))
        o = np.zeros(1500, np.uint8)
        if name == "fx":
            last = None
            for f in range(W + K):
                x = np.ascontiguousarray(base[0, f * n * ch:(f + 1) * n * ch]); l = R.opus_encode(enc, x.ctypes.data_as(ctypes.c_void_p), n, o.ctypes.data_as(ctypes.c_void_p), 1276)
                r = ctypes.c_uint32(0); R.opus_encoder_ctl(enc, 4031, ctypes.byref(r)); last = (l, r.value, bytes(o[:l]))
            res["matches_reference"] = bool(last[0] == int(lens[0]) and last[1] == int(rng[0]) and last[2] == bytes(out[0, :last[0]]) and (lens[::U] == lens[0]).all())
        else:
            nf = 0; t0 = time.perf_counter()
            while a.cpu_seconds > 0 and time.perf_counter() - t0 < a.cpu_seconds:
                for f in range(W + K):
                    x = np.ascontiguousarray(base[0, f * n * ch:(f + 1) * n * ch]); R.opus_encode(enc, x.ctypes.data_as(ctypes.c_void_p), n, o.ctypes.data_as(ctypes.c_void_p), 1276); nf += 1
            dt = time.perf_counter() - t0
            if nf == 0: R.opus_encoder_destroy(enc); continue
            res["cpu_baseline"] = {"value": nf / dt, "unit": "frames/s", "cores": 1, "kind": "reference", "sample": "%d consecutive 20 ms frames of stream 0's signal, libopus float build (RTCD), one thread" % nf}
            res["speedup_vs_one_core"] = res["value"] / (nf / dt)
        R.opus_encoder_destroy(enc)
    b.close()
    print(json.dumps(res))
if __name__ == "__main__":
    main()
