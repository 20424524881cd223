"""tools/silk_bench.py — throughput of the batched SILK noise-shaping quantiser (config 3's dominant kernel: voip 16 kHz mono 20 ms,
complexity 10 -> NSQ_del_dec with 4 states, shaping order 24, warping on) with inputs resident in HBM, HIP-event timed, next to the
compiled reference's silk_NSQ_del_dec_c on one host core.  Prints one JSON line.  (Not the round's headline bench: that is bench.py.)"""
import argparse, ctypes, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=65536); ap.add_argument("--steps", type=int, default=5); ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--states", type=int, default=4); ap.add_argument("--plain", action="store_true", help="plain NSQ (1 state, no warping)")
    ap.add_argument("--cpu-frames", type=int, default=4000)
    ap.add_argument("--kernel", default="nsq", choices=["nsq", "resampler", "lpc", "pitch", "decode"])
    a = ap.parse_args()
    import torch, opus_amd
    if a.kernel != "nsq": return hbm_kernels(a, torch, opus_amd)
    from silk_inputs import NSQ_FRAME, make_cfg, make_frame, make_input, fresh_state
    cfg = make_cfg(16, 4, 24, 1 if a.plain else a.states, not a.plain)
    dd = not a.plain
    n = a.streams; L = 320
    rng = np.random.default_rng(1)
    U = 512                                            # unique parameter sets, tiled over the batch
    fr_u = np.array([make_frame(rng, cfg) for _ in range(U)], dtype=NSQ_FRAME)
    x_u = np.stack([make_input(rng, cfg, fr_u[s]["Gains_Q16"]) for s in range(U)])
    idx = np.arange(n) % U
    fr = fr_u[idx]; x = x_u[idx]
    dev = torch.device("cuda:0")
    d_fr = torch.from_numpy(fr.view(np.uint8).reshape(n, -1).copy()).to(dev); d_x = torch.from_numpy(x).to(dev)
    d_p = torch.zeros((n, L), dtype=torch.int8, device=dev)
    b = opus_amd.NsqBatch(n, cfg)
    torch.cuda.synchronize()
    for _ in range(a.warmup): b.run_dev(d_fr.data_ptr(), d_x.data_ptr(), d_p.data_ptr())
    b.sync()
    ms = b.time_dev(d_fr.data_ptr(), d_x.data_ptr(), d_p.data_ptr(), a.steps)
    fps = n * a.steps / (ms * 1e-3)
    # spot check against the oracle: after warmup+steps frames of identical input per stream the last pulses of stream s equal the oracle's
    out = {"kernel": "silk_nsq_del_dec" if dd else "silk_nsq", "streams": n, "steps": a.steps, "ms_per_step": ms / a.steps, "frames_per_s": fps,
           "samples_per_s": fps * L, "config": {"fs_kHz": 16, "nb_subfr": 4, "states": int(cfg[4]), "shaping": 24, "warping_Q16": int(cfg[5])}}
    # CPU baseline: the compiled reference (fixed-point build, C path) on one core
    from reflib import ref_expose
    X = ref_expose()
    if X is not None and a.cpu_frames > 0:
        from test_oracle_silk import run_ref
        st = fresh_state(U)
        t0 = time.perf_counter(); k = 0
        while k < a.cpu_frames:
            s = k % U
            run_ref(cfg, dd, st[s:s + 1], fr_u[s], x_u[s]); k += 1
        dt = time.perf_counter() - t0
        # subtract the ctypes marshalling overhead measured with a no-op-sized call? keep it simple: report as is, it is ~15 us/call
        out["cpu_baseline"] = {"value": a.cpu_frames / dt, "unit": "frames/s", "cores": 1, "kind": "reference", "sample": "%d frames, silk_NSQ%s_c via ctypes (incl. ~10 us/call marshalling)" % (a.cpu_frames, "_del_dec" if dd else "")}
        out["speedup_vs_1core"] = fps / (a.cpu_frames / dt)
    print(json.dumps(out))

def hbm_kernels(a, torch, opus_amd):
    """the two HBM-bound SILK kernels: resampler 48 -> 16 kHz (20 ms per channel per step) and the order-16 LPC analysis filter over the
    672-sample pitch-analysis buffer (silk/fixed/find_pitch_lags_FIX.c); achieved GB/s = algorithmic bytes / HIP-event time."""
    import ctypes
    from reflib import ref_expose, oracle
    n = a.streams; dev = torch.device("cuda:0"); rng = np.random.default_rng(2)
    stream = torch.cuda.current_stream().cuda_stream
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if a.kernel == "decode":
        # SILK-only (WB, 20 ms, mono) and hybrid (FB, 20 ms, mono) packets from the compiled reference encoder, U distinct streams tiled over the batch
        from test_kernel_emu_silkdec import speechy
        from test_oracle_encoder import RefEnc
        from test_oracle_decoder import RefDec
        out = {"kernel": "oa_decode_kernel on SILK-only / hybrid packets", "streams": n}
        for name, mode, bw, br in (("silk_wb", 1000, 1103, 20000), ("hybrid_fb", 1001, 1105, 32000)):
            U = 16; F = a.steps + a.warmup
            pk = []
            for u in range(U):
                sig = speechy(F, 1, 300 + u, 960); e = RefEnc(1, application=2048, force_mode=mode, bandwidth=bw, bitrate=br)
                pk.append([e.encode(np.ascontiguousarray(sig[i * 960:(i + 1) * 960]), 960)[0] for i in range(F)])
            stride = 256
            buf = np.zeros((F, U, stride), np.uint8); lens = np.zeros((F, U), np.int32)
            for f in range(F):
                for u in range(U): buf[f, u, :len(pk[u][f])] = np.frombuffer(pk[u][f], np.uint8); lens[f, u] = len(pk[u][f])
            idx = np.arange(n) % U
            d_buf = [torch.from_numpy(buf[f][idx]).to(dev) for f in range(F)]; d_len = [torch.from_numpy(lens[f][idx]).to(dev) for f in range(F)]
            d_pcm = torch.zeros((n, 960), dtype=torch.int16, device=dev); d_ns = torch.zeros(n, dtype=torch.int32, device=dev); d_rng = torch.zeros(n, dtype=torch.int32, device=dev)
            b = opus_amd.DecoderBatch(n, channels=1)
            for f in range(a.warmup): b.decode_dev(d_buf[f].data_ptr(), stride, d_len[f].data_ptr(), d_pcm.data_ptr(), 960, d_ns.data_ptr(), d_rng.data_ptr(), stream)
            torch.cuda.synchronize(); e0.record()
            for f in range(a.warmup, F): b.decode_dev(d_buf[f].data_ptr(), stride, d_len[f].data_ptr(), d_pcm.data_ptr(), 960, d_ns.data_ptr(), d_rng.data_ptr(), stream)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / a.steps
            ok = bool((d_ns.cpu().numpy() == 960).all())
            # spot check of the last step against the compiled reference decoder
            r = RefDec(1); ref = None
            for f in range(F): ref = r.decode(pk[0][f])
            same = bool(np.array_equal(d_pcm[0].cpu().numpy(), ref[1][:, 0]))
            out[name] = {"ms_per_step": ms, "frames_per_s": n / (ms * 1e-3), "all_frames_ok": ok, "matches_reference": same, "mean_packet_bytes": float(lens.mean())}
            if a.cpu_frames > 0:
                r = RefDec(1); t0 = time.perf_counter(); k = 0
                while k < a.cpu_frames: r.decode(pk[0][k % F]); k += 1
                out[name]["cpu_baseline"] = {"value": a.cpu_frames / (time.perf_counter() - t0), "unit": "frames/s", "cores": 1, "kind": "reference", "sample": "%d x opus_decode via ctypes" % a.cpu_frames}
            b.close()
        print(json.dumps(out)); return
    if a.kernel == "pitch":
        from silk_inputs import make_pitch_frame
        from test_kernel_emu_silk import PE_IN, PE_OUT
        U = 256; xu = np.stack([make_pitch_frame(rng, 16, 4, "voiced" if i % 4 else "noise")[0] for i in range(U)])
        x = torch.from_numpy(xu[np.arange(n) % U]).to(dev)
        pin = np.zeros(n, PE_IN); pin["search_thres1_Q16"] = int(0.7 * 65536); pin["search_thres2_Q13"] = int(0.3 * 8192); pin["prevLag"] = 120; pin["LTPCorr_Q15"] = 16000
        d_in = torch.from_numpy(pin.view(np.uint8).reshape(n, -1).copy()).to(dev); d_out = torch.zeros((n, 24), dtype=torch.uint8, device=dev)
        Lb = opus_amd.lib()
        for _ in range(a.warmup): Lb.opusgpu_silk_pitch_analysis_batch_dev(0, n, x.data_ptr(), d_in.data_ptr(), d_out.data_ptr(), 16, 2, 4, stream)
        torch.cuda.synchronize(); e0.record()
        for _ in range(a.steps): Lb.opusgpu_silk_pitch_analysis_batch_dev(0, n, x.data_ptr(), d_in.data_ptr(), d_out.data_ptr(), 16, 2, 4, stream)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.steps; bytes_per = 640 * 2 + 16 + 24
        voiced = int((np.frombuffer(d_out.cpu().numpy().tobytes(), dtype=PE_OUT)["unvoiced"] == 0).sum())
        out = {"kernel": "silk_pitch_analysis_core 16 kHz, 20 ms, complexity 2", "frames": n, "voiced": voiced, "ms_per_step": ms, "frames_per_s": n / (ms * 1e-3), "bytes_per_frame": bytes_per,
               "achieved_GBps": n * bytes_per / (ms * 1e-3) / 1e9, "hbm_peak_GBps": 8000}
        from reflib import ref_fx
        R = ref_fx()
        if R is not None and a.cpu_frames > 0:
            pitch = np.zeros(4, np.int32); li = np.zeros(1, np.int16); ci = np.zeros(1, np.int8)
            vp = lambda z: z.ctypes.data_as(ctypes.c_void_p)
            t0 = time.perf_counter()
            for k in range(a.cpu_frames):
                lc = np.array([16000], np.int32)
                R.silk_pitch_analysis_core(vp(xu[k % U]), vp(pitch), vp(li), vp(ci), vp(lc), 120, int(0.7 * 65536), int(0.3 * 8192), 16, 2, 4, 0)
            dt = time.perf_counter() - t0
            out["cpu_baseline"] = {"value": a.cpu_frames / dt, "unit": "frames/s", "cores": 1, "kind": "reference", "sample": "%d x silk_pitch_analysis_core via ctypes" % a.cpu_frames}
    elif a.kernel == "resampler":
        x = torch.from_numpy(rng.integers(-20000, 20000, (n, 960)).astype(np.int16)).to(dev); y = torch.zeros((n, 320), dtype=torch.int16, device=dev)
        b = opus_amd.ResamplerBatch(n, 48000, 16000, 1)
        for _ in range(a.warmup): b.run_dev(y.data_ptr(), x.data_ptr(), 960, stream)
        torch.cuda.synchronize(); e0.record()
        for _ in range(a.steps): b.run_dev(y.data_ptr(), x.data_ptr(), 960, stream)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.steps; bytes_per = 960 * 2 + 320 * 2 + 2 * 90 * 4
        out = {"kernel": "silk_resampler 48k->16k", "channels": n, "ms_per_step": ms, "frames_per_s": n / (ms * 1e-3), "bytes_per_frame": bytes_per,
               "achieved_GBps": n * bytes_per / (ms * 1e-3) / 1e9, "hbm_peak_GBps": 8000}
        X = ref_expose()
        if X is not None and a.cpu_frames > 0:
            st = np.zeros(X.ref_silk_resampler_state_size(), np.uint8); X.ref_silk_resampler_init(st.ctypes.data_as(ctypes.c_void_p), 48000, 16000, 1)
            xi = rng.integers(-20000, 20000, 960).astype(np.int16); yo = np.zeros(320, np.int16)
            t0 = time.perf_counter()
            for _ in range(a.cpu_frames): X.ref_silk_resampler(st.ctypes.data_as(ctypes.c_void_p), yo.ctypes.data_as(ctypes.c_void_p), xi.ctypes.data_as(ctypes.c_void_p), 960)
            dt = time.perf_counter() - t0
            out["cpu_baseline"] = {"value": a.cpu_frames / dt, "unit": "frames/s", "cores": 1, "kind": "reference", "sample": "%d x silk_resampler(960 samples) via ctypes" % a.cpu_frames}
    else:
        L, d = 672, 16
        x = torch.from_numpy(rng.integers(-20000, 20000, (n, L)).astype(np.int16)).to(dev); y = torch.zeros((n, L), dtype=torch.int16, device=dev)
        B = torch.from_numpy(rng.integers(-3000, 3000, (n, d)).astype(np.int16)).to(dev)
        Lb = opus_amd.lib()
        for _ in range(a.warmup): Lb.opusgpu_silk_lpc_analysis_filter_batch_dev(0, n, y.data_ptr(), x.data_ptr(), B.data_ptr(), L, d, stream)
        torch.cuda.synchronize(); e0.record()
        for _ in range(a.steps): Lb.opusgpu_silk_lpc_analysis_filter_batch_dev(0, n, y.data_ptr(), x.data_ptr(), B.data_ptr(), L, d, stream)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.steps; bytes_per = 4 * L + 2 * d
        out = {"kernel": "silk_lpc_analysis_filter len=672 d=16", "signals": n, "ms_per_step": ms, "signals_per_s": n / (ms * 1e-3), "bytes_per_signal": bytes_per,
               "achieved_GBps": n * bytes_per / (ms * 1e-3) / 1e9, "hbm_peak_GBps": 8000}
        X = ref_expose()
        if X is not None and a.cpu_frames > 0:
            xi = rng.integers(-20000, 20000, L).astype(np.int16); yo = np.zeros(L, np.int16); Bi = rng.integers(-3000, 3000, d).astype(np.int16)
            t0 = time.perf_counter()
            for _ in range(a.cpu_frames): X.ref_silk_lpc_analysis_filter(yo.ctypes.data_as(ctypes.c_void_p), xi.ctypes.data_as(ctypes.c_void_p), Bi.ctypes.data_as(ctypes.c_void_p), L, d)
            dt = time.perf_counter() - t0
            out["cpu_baseline"] = {"value": a.cpu_frames / dt, "unit": "signals/s", "cores": 1, "kind": "reference", "sample": "%d x silk_LPC_analysis_filter(672, 16) via ctypes" % a.cpu_frames}
    print(json.dumps(out))

if __name__ == "__main__":
    main()
