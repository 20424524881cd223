#!/usr/bin/env python3
"""tools/parity_soak.py — the parity gate of SURVEY §8d, run on the GPU box: for BASELINE configs 2, 3 and 4, S sampled streams x T consecutive
frame-steps through the C ABI batch encoder (state carried on the device) against the compiled reference's opus_encode (oracle/_ref, fixed-point build) on
the same PCM: packet bytes, lengths and OPUS_GET_FINAL_RANGE must agree for every (stream, frame).  One JSON line per config.  Test infrastructure."""
import argparse, ctypes, json, os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
CONFIGS = {
    2: dict(name="config 2: CELT-only, restricted-lowdelay, 48 kHz stereo, 20 ms, CVBR 128 kb/s, complexity 10", Fs=48000, ch=2, app=2051, ctl=((4002, 128000), (4010, 10)), sig="music"),
    3: dict(name="config 3: SILK-only, VOIP 16 kHz mono WB, 20 ms, 24 kb/s, complexity 10", Fs=16000, ch=1, app=2048, ctl=((11002, 1000), (4008, 1103), (4002, 24000), (4010, 10)), sig="speech"),
    4: dict(name="config 4: hybrid, AUDIO 48 kHz stereo FB, 20 ms, VBR 128 kb/s, complexity 10", Fs=48000, ch=2, app=2049, ctl=((11002, 1001), (4008, 1105), (4002, 128000), (4010, 10)), sig="speech"),
}
# mid-stream control changes applied to every stream of the batch and to every reference encoder at the same frame: (frame, request, value)
SCHEDULE = {
    2: [(150, 4002, 96000), (300, 4010, 5), (450, 4006, 0), (600, 4006, 1), (600, 4020, 0), (750, 4002, 160000), (750, 4010, 10), (900, 4022, 1)],
    3: [(150, 4002, 16000), (300, 4010, 4), (450, 4012, 1), (450, 4014, 15), (600, 4016, 1), (750, 4002, 32000), (900, 4010, 10)],
    4: [(150, 4002, 96000), (300, 4010, 6), (450, 4008, 1104), (600, 4012, 1), (600, 4014, 10), (750, 4002, 160000), (750, 4008, 1105), (900, 4010, 10)],
}
def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--ctl-schedule", action="store_true", help="apply SCHEDULE[config] while the streams run"); ap.add_argument("--streams", type=int, default=256); ap.add_argument("--frames", type=int, default=1000); ap.add_argument("--configs", default="2,3,4"); ap.add_argument("--bases", type=int, default=32)
    ap.add_argument("--float-analysis", action="store_true", help="the encoder with its tonality / music analysis (the library default) against the reference built WITH the float API (libopus_ref_fxa.so)")
    a = ap.parse_args()
    import opus_amd, signals
    from reflib import ref_fx, ref_fxa
    from silk_enc_bench import speech
    R = ref_fxa() if a.float_analysis else ref_fx(); assert R is not None, "compiled reference (oracle/_ref) missing"
    R.opus_encoder_create.restype = ctypes.c_void_p
    for cid in [int(x) for x in a.configs.split(",")]:
        c = CONFIGS[cid]; Fs, ch, n = c["Fs"], c["ch"], c["Fs"] // 50
        S, T, U = a.streams, a.frames, min(a.bases, a.streams)
        t0 = time.time()
        if c["sig"] == "music": base = [signals.music(T + 1, channels=ch, seed=1000 + u).reshape(-1, ch)[: (T + 1) * n] for u in range(U)]
        else: base = [np.stack([speech(Fs, (T + 1) * n, 100 + u + 800 * k) for k in range(ch)], 1) for u in range(U)]
        def stream(s):
            x = np.roll(base[s % U], (s // U) * 4801, axis=0)[: T * n].astype(np.float64) * (1.0 - 0.07 * (s // U))
            return np.ascontiguousarray(x.astype(np.int16))
        sig = [stream(s) for s in range(S)]
        b = opus_amd.EncoderBatch(S, channels=ch, application=c["app"], Fs=Fs)
        for req, v in c["ctl"]: b.ctl(req, v)
        b.ctl(opus_amd.OPUS_AMD_SET_FLOAT_ANALYSIS_REQUEST, 1 if a.float_analysis else 0)
        gp = [[None] * T for _ in range(S)]; gr = np.zeros((S, T), np.uint32)
        sched = SCHEDULE[cid] if a.ctl_schedule else []
        for f in range(T):
            for (ff, req, v) in sched:
                if ff == f: b.ctl(req, v)
            pcm = np.stack([sig[s][f * n:(f + 1) * n].reshape(-1) for s in range(S)])
            pk, lens, rng = b.encode(pcm, n)
            for s in range(S): gp[s][f] = pk[s] if int(lens[s]) > 0 else int(lens[s])
            gr[:, f] = np.asarray(rng, dtype=np.uint32)
        b.close()
        t_gpu = time.time() - t0; t0 = time.time()
        bad = 0; first = None; nbytes = 0
        for s in range(S):
            err = ctypes.c_int(0); enc = ctypes.c_void_p(R.opus_encoder_create(Fs, ch, c["app"], ctypes.byref(err)))
            for req, v in c["ctl"]: R.opus_encoder_ctl(enc, req, ctypes.c_int(v))
            o = np.zeros(1500, np.uint8); r = ctypes.c_uint32(0)
            for f in range(T):
                for (ff, req, v) in sched:
                    if ff == f: assert R.opus_encoder_ctl(enc, req, ctypes.c_int(v)) == 0
                x = sig[s][f * n:(f + 1) * n]
                l = R.opus_encode(enc, x.ctypes.data_as(ctypes.c_void_p), n, o.ctypes.data_as(ctypes.c_void_p), 1276)
                R.opus_encoder_ctl(enc, 4031, ctypes.byref(r))
                nbytes += max(l, 0)
                if not (isinstance(gp[s][f], bytes) and len(gp[s][f]) == l and gp[s][f] == bytes(o[:l]) and int(gr[s, f]) == r.value):
                    bad += 1
                    if first is None: first = [s, f, l, len(gp[s][f]) if isinstance(gp[s][f], bytes) else gp[s][f]]
            R.opus_encoder_destroy(enc)
        print(json.dumps({"parity_gate": c["name"], "streams": S, "frames_per_stream": T, "stream_frames_checked": S * T, "mismatches": bad, "first_mismatch": first,
                          "mean_packet_bytes": nbytes / (S * T), "reference_build": "FIXED_POINT + float API (analysis.c, mlp.c)" if a.float_analysis else "FIXED_POINT + DISABLE_FLOAT_API", "distinct_base_signals": U, "ctl_changes": len(sched), "gpu_seconds_incl_host_copies": round(t_gpu, 1), "reference_seconds_one_core": round(time.time() - t0, 1)}), flush=True)
if __name__ == "__main__": main()
