#!/usr/bin/env python3
"""bench.py — encoded frames/s of the batched Opus encoder on N MI355X (one process per GPU).

Headline workload (default, --config 2) = BASELINE.json configs[1]: CELT-only encode, OPUS_APPLICATION_RESTRICTED_LOWDELAY, 48 kHz stereo, 20 ms
frames, 128 kb/s CVBR, complexity 10, 65,536 independent streams per GPU; a "step" = one 20 ms frame-step of every stream (65,536 frames per GPU), state
carried in HBM between steps, PCM resident in HBM before the timed region.  --config 3 (SILK-only VOIP 16 kHz mono 24 kb/s), 4 (hybrid AUDIO 48 kHz stereo
128 kb/s VBR, the 8-GPU configuration of BASELINE.json: 65,536 streams per GPU) and 5 (multistream: 257 encoders x 255 mono AUDIO streams, one batch) run the
other BASELINE configurations through the same harness.  At N = 1 the default run also times configs 3 and 4 briefly and reports them in "configs".
Streams shard across ranks with no data-path collective; the only exchange is the final gather of the compacted packets to rank 0 over RCCL, included in the
timed region when N > 1 (weak scaling: per-GPU work fixed).

Prints ONE JSON line (rank 0): metric/value + "roofline" (HBM-bound: algorithmic bytes / kernel time measured live with HIP events on the launch stream; peak = the
8 TB/s of the guide, peak_measured = a device-to-device copy timed in this run) and "cpu_baseline" (the compiled reference on one pinned host core, on the PCM of
stream 0 copied back from the GPU batch; plus an all-cores figure and the host's core count / CPU model).
"""
import argparse, ctypes, json, os, sys, time
import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

NO_ANALYSIS = False      # --no-analysis
CORPUS = "reference"     # 48 kHz configurations: the reference's generate_music() tunes (SURVEY.md 8d names them first); --corpus pool: this repo's music / noise-burst pool
CPU_STREAMS = 64         # streams of the GPU batch the CPU legs (baseline, parity sample) run
GATHER = "auto"          # --gather
CONFIGS = {
    2: dict(name="CELT-only encode, restricted-lowdelay, 48 kHz stereo, 20 ms, CVBR 128 kb/s, complexity 10", app=2051, Fs=48000, ch=2, kernel="oa_encode_kernel",
            ctls=((4002, 128000), (4010, 10)), metric="encoded frames/s (48 kHz stereo, 20 ms, complexity 10)"),
    3: dict(name="SILK-only encode, VOIP, 16 kHz mono, 20 ms, wideband, VBR 24 kb/s, complexity 10", app=2048, Fs=16000, ch=1, kernel="oa_sh_front_kernel + the pred stage's four kernels + oa_sh_quant_kernel + oa_sh_back_kernel (one call)",
            ctls=((11002, 1000), (4008, 1103), (4002, 24000), (4010, 10)), metric="encoded frames/s (SILK-only, 16 kHz mono, 20 ms, complexity 10)"),
    4: dict(name="hybrid encode, AUDIO, 48 kHz stereo, 20 ms, fullband, VBR 128 kb/s, complexity 10", app=2049, Fs=48000, ch=2, kernel="oa_sh_front_kernel + the pred stage's four kernels + oa_sh_quant_kernel + oa_sh_back_kernel (one call)",
            ctls=((11002, 1001), (4008, 1105), (4006, 1), (4002, 128000), (4010, 10)), metric="encoded frames/s (hybrid, 48 kHz stereo, 20 ms, complexity 10)"),
    # what a VoIP deployment of config 3 also runs (not BASELINE.json rows; "extra" legs of the default line): in-band FEC at 10 % expected loss, and 60 ms packets
    31: dict(name="config 3 + OPUS_SET_INBAND_FEC(1), OPUS_SET_PACKET_LOSS_PERC(10)", app=2048, Fs=16000, ch=1, kernel="the SILK-capable pipeline (the LBRR pass in the quantiser kernel)", key="config_3_fec",
             ctls=((11002, 1000), (4008, 1103), (4002, 24000), (4010, 10), (4012, 1), (4014, 10)), metric="encoded frames/s (SILK-only + in-band FEC, 16 kHz mono, 20 ms, complexity 10)"),
    32: dict(name="config 3 in 60 ms packets (three SILK frames per call)", app=2048, Fs=16000, ch=1, frame_ms=60, kernel="the SILK-capable pipeline, its front -> pred -> quantiser relay once per 20 ms frame", key="config_3_60ms",
             ctls=((11002, 1000), (4008, 1103), (4002, 24000), (4010, 10)), metric="encoded 60 ms packets/s (SILK-only, 16 kHz mono, complexity 10); x 3 = 20 ms frames/s"),
    5: dict(name="multistream, 255 mono AUDIO streams per encoder (mapping family 255), 48 kHz, 20 ms, 64 kb/s per stream, complexity 10; 257 encoders = 65,535 elementary streams",
            app=2049, Fs=48000, ch=1, kernel="oa_sh_front_kernel + the pred stage's four kernels + oa_sh_quant_kernel + oa_sh_back_kernel", ctls=((4002, 64000), (4010, 10)), metric="encoded elementary-stream frames/s (255-channel multistream, 48 kHz, 20 ms, complexity 10)"),
}

def frame_of(cfg): return cfg["Fs"] * cfg.get("frame_ms", 20) // 1000          # samples per channel of one call

def reference_music(nsamp, seeds, starts=None):
    """the reference's own test corpus: generate_music() of tests/test_opus_encode.c:57-85 (a byte-beat tune through two rounding one-pole filters, dithered with the
    fast_rand() multiply-with-carry generator of tests/test_opus_common.h:56-62), restated here because the bench input must not come from oracle/.  One tune per seed
    (Rz = Rw = seed, as the test program seeds them), all tunes advanced together: the recurrence is serial in time but independent across tunes, so every step is one
    numpy operation over the seed axis.  int16 [len(seeds), nsamp, 2].
    starts[p] (default 2880: right after the 60 ms of leading silence, which a bench step should not time) = the sample of the 30 s piece tune p begins at: the melody counter
    j is a function of the sample index alone (it steps at every i % 6 == 0 from 2880 on), so a tune can be entered anywhere -- filters and dither restart there (their memory
    is a few dozen samples) -- and a pool of tunes covers the whole piece instead of its first quarter second.  starts = 2880 reproduces the C function bit for bit."""
    n = len(seeds)
    z0 = np.array(seeds, np.uint64) & np.uint64(0xffffffff); z0[z0 == 0] = 1
    i0 = np.full(n, 2880, np.int64) if starts is None else np.asarray(starts, np.int64)
    # fast_rand()'s two multiply-with-carry generators, all 2 * nsamp draws of every tune at once: z' = a (z & 65535) + (z >> 16) is z' = z * 65536^-1 mod (65536 a - 1)
    # (z' * 65536 = z + (z & 65535)(65536 a - 1)), so draw t is z0 * 65536^-t mod m -- a table of powers, no loop over time
    def mwc(a_):
        m = a_ * 65536 - 1; binv = pow(65536, -1, m); C = 4096; Q = (2 * nsamp + C) // C + 1
        def powers(base, cnt):
            o = np.ones(cnt, np.uint64); k = 1
            while k < cnt:
                o[k:2 * k] = (o[:min(k, cnt - k)] * np.uint64(pow(base, k, m))) % np.uint64(m); k *= 2
            return o
        pw = ((powers(pow(binv, C, m), Q)[:, None] * powers(binv, C)[None, :]) % np.uint64(m)).reshape(-1)[1:2 * nsamp + 1]
        return lambda zz: (zz[:, None] * pw[None, :]) % np.uint64(m)
    draw_z, draw_w = mwc(36969), mwc(18000)
    k = np.arange(nsamp, dtype=np.int64)
    U = np.empty((nsamp, n, 2), np.int32)                                            # v - a, a = the previous v (0 before the first sample), of every tune
    for p0 in range(0, n, 8):                                                        # (8 tunes at a time: the draw tables are [tunes, 2 * nsamp] uint64)
        sl = slice(p0, min(n, p0 + 8))
        r = ((draw_z(z0[sl]) << np.uint64(16)) + draw_w(z0[sl])) & np.uint64(0xffffffff)
        dith = (r & np.uint64(65535)).astype(np.int64) - (r >> np.uint64(16)).astype(np.int64)              # [q, 2 * nsamp]: v1's and v2's dither, interleaved
        j = ((i0[sl] + 5) // 6 - 480)[:, None] + ((i0[sl][:, None] + k[None, :] + 5) // 6 - ((i0[sl] + 5) // 6)[:, None])   # the melody counter at every sample (it steps after each i % 6 == 0)
        v = (((j * ((j >> 12) ^ ((j >> 10 | j >> 12) & 26 & j >> 7))) & 128) + 128) << 15
        vv = np.stack([v + dith[:, 0::2], v + dith[:, 1::2]], 2)                    # [q, nsamp, 2]
        vv[:, 1:] -= vv[:, :-1].copy()
        U[:, sl] = vv.transpose(1, 0, 2)
    # the two rounding one-pole filters are the only serial part: one pass over time for all tunes and both channels
    bb = np.zeros((n, 2), np.int64); cc = np.zeros((n, 2), np.int64); res = np.empty((nsamp, n, 2), np.int32)
    for t in range(nsamp):
        bn = U[t] + ((bb * 61 + 32) >> 6)
        cc = (30 * (cc + bn + bb) + 32) >> 6
        bb = bn
        res[t] = cc
    return np.clip((res.astype(np.int64) + 128) >> 8, -32768, 32767).astype(np.int16).transpose(1, 0, 2).copy()

def synth(cfg, T, n_pool, rank, corpus="pool"):
    """pool of distinct signals [n_pool, (T+2)*frame*ch] int16 at the config's rate"""
    import signals
    Fs, ch, fr = cfg["Fs"], cfg["ch"], frame_of(cfg)
    if corpus == "reference" and Fs == 48000:
        span = 30 * 48000 - 2880 - (T + 2) * fr                                                      # the piece is 30 s long in the reference's test (SAMPLES, tests/test_opus_encode.c:51)
        tunes = reference_music((T + 2) * fr, [42 + 100000 * rank + p for p in range(n_pool)],      # Rz = Rw = seed0 + stream, seed0 = 42 (one tune per pool slot)
                                starts=[2880 + (p * span) // n_pool for p in range(n_pool)])           # ... the pool's tunes enter the piece at evenly spread points
        return (tunes if ch == 2 else tunes[:, :, :1]).reshape(n_pool, -1)
    if Fs == 48000:
        return np.stack([(signals.music(T + 2, channels=ch, seed=1000 * rank + p) if p % 4 else signals.noise_bursts(T + 2, channels=ch, seed=1000 * rank + p)).reshape(-1) for p in range(n_pool)])
    rng = np.random.default_rng(77 + rank)
    out = []
    for p in range(n_pool):                                    # glottal-like harmonic source, gated, with noise
        n = (T + 2) * fr; t = np.arange(n) / Fs
        f0 = 120 + 30 * np.sin(2 * np.pi * 0.7 * t + p) + (p % 7) * 9
        ph = 2 * np.pi * np.cumsum(f0) / Fs
        s = sum(np.sin(k * ph) / k for k in range(1, 25)) * (np.sin(2 * np.pi * 1.5 * t + p) > -0.3) * 6000 + rng.normal(0, 120, n)
        s = np.clip(s, -32768, 32767).astype(np.int16)
        out.append(np.repeat(s[:, None], ch, 1).reshape(-1) if ch > 1 else s)
    return np.stack(out)

def host_info():
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"): model = line.split(":", 1)[1].strip(); break
    except Exception: pass
    try: ncpu = len(os.sched_getaffinity(0))
    except Exception: ncpu = os.cpu_count() or 1
    return model, ncpu

def _ref_lib(path):
    L = ctypes.CDLL(path, mode=ctypes.RTLD_LOCAL)
    vp = ctypes.c_void_p
    L.opus_encoder_create.restype = vp; L.opus_encoder_create.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    L.opus_encode.argtypes = [vp, vp, ctypes.c_int, vp, ctypes.c_int]
    L.opus_encoder_ctl.argtypes = [vp, ctypes.c_int, ctypes.c_int]
    L.opus_encoder_destroy.argtypes = [vp]; L.opus_encoder_destroy.restype = None
    L.opus_decoder_create.restype = vp; L.opus_decoder_create.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    L.opus_decode.argtypes = [vp, vp, ctypes.c_int, vp, ctypes.c_int, ctypes.c_int]
    L.opus_decoder_destroy.argtypes = [vp]; L.opus_decoder_destroy.restype = None
    return L

def _cpu_worker(args):
    """One worker of a CPU leg: ONE codec state, created and destroyed outside the clock, living through consecutive frames (>= 3,000 of them) -- the sample's
    frames in order (stream 0's frames, then stream 1's, ...: the encoder meets the joins as it would meet any cut in its input), the whole sample again when it runs out, until
    at least `min_frames` frames and `seconds` have passed.  kind "enc": pcm [n_streams, T, frame*ch] int16; kind "dec": (packets [T, n_streams, stride] uint8, lens [T, n_streams]).
    Returns (frames/s of this worker, frames it processed)."""
    path, cfg, data, seconds, pin, kind, min_frames = args
    if pin is not None:
        try: os.sched_setaffinity(0, {pin})
        except Exception: pass
    L = _ref_lib(path)
    err = ctypes.c_int()
    fr = frame_of(cfg)
    n = 0; done = False
    if kind == "enc":
        pcm = data; ns, T = pcm.shape[0], pcm.shape[1]
        out = (ctypes.c_ubyte * 1500)()
        st = L.opus_encoder_create(cfg["Fs"], cfg["ch"], cfg["app"], ctypes.byref(err))
        for req, v in cfg["ctls"]: L.opus_encoder_ctl(st, req, v)
        ptrs = [pcm[s_, i].ctypes.data for s_ in range(ns) for i in range(T)]
        t0 = time.perf_counter()
        while not done:
            for p_ in ptrs: L.opus_encode(st, p_, fr, out, 1276)
            n += len(ptrs)
            done = n >= min_frames and time.perf_counter() - t0 > seconds
        dt = time.perf_counter() - t0
        L.opus_encoder_destroy(st)
    else:
        pk, lens = data; T, ns = lens.shape
        pcm = (ctypes.c_int16 * (fr * cfg["ch"]))()
        st = L.opus_decoder_create(cfg["Fs"], cfg["ch"], ctypes.byref(err))
        items = [(pk[i, s_].ctypes.data, int(lens[i, s_])) for s_ in range(ns) for i in range(T)]
        t0 = time.perf_counter()
        while not done:
            for p_, l_ in items: L.opus_decode(st, p_, l_, pcm, fr, 0)
            n += len(items)
            done = n >= min_frames and time.perf_counter() - t0 > seconds
        dt = time.perf_counter() - t0
        L.opus_decoder_destroy(st)
    return n / dt, n

CPU_MIN_FRAMES = 3000    # consecutive frames a CPU leg's codec state lives through at least
def cpu_baseline(cfg, data, seconds=10.0, all_cores_seconds=4.0, kind="enc"):
    """Reference libopus (default float build, RTCD/AVX2) on ONE pinned host core over a sample of the GPU batch copied back from the device: the first CPU_STREAMS
    streams, every frame the GPU batch saw of them, through ONE codec state that lives through >= CPU_MIN_FRAMES consecutive frames (create / destroy outside the clock);
    then one worker per host core for the all-cores figure.  The strings that describe the legs are the same for every leg: NOTES (printed once, under "notes")."""
    path = os.path.join(ROOT, "oracle/_ref/libopus_ref_fl.so")
    model, ncpu = host_info()
    if not os.path.exists(path):
        return {"value": None, "unit": "frames/s", "cores": 1, "kind": "reference", "sample": "oracle/_ref/libopus_ref_fl.so did not travel"}
    try: cpus = sorted(os.sched_getaffinity(0)); first = cpus[0]
    except Exception: cpus = list(range(ncpu)); first = None
    one, nfr = _cpu_worker((path, cfg, data, seconds, first, kind, CPU_MIN_FRAMES))
    try: os.sched_setaffinity(0, set(cpus))                                 # the single-core leg pinned this process: undo before spawning the pool
    except Exception: pass
    allc = None
    if all_cores_seconds > 0 and ncpu > 1:
        try:
            import multiprocessing as mp
            with mp.get_context("spawn").Pool(len(cpus)) as pool:            # (never fork a process that holds a HIP context)
                allc = float(sum(r[0] for r in pool.map(_cpu_worker, [(path, cfg, data, all_cores_seconds, c, kind, 0) for c in cpus])))
        except Exception:
            allc = None
    same = None                                                              # the library the GPU path is bit-exact to: fixed-point arithmetic (+ the float analysis unless --no-analysis)
    try:
        fx = os.path.join(ROOT, "oracle/_ref/libopus_ref_fx.so" if NO_ANALYSIS else "oracle/_ref/libopus_ref_fxa.so")
        if os.path.exists(fx): same = round(float(_cpu_worker((fx, cfg, data, min(4.0, seconds), first, kind, CPU_MIN_FRAMES))[0]), 1)
        try: os.sched_setaffinity(0, set(cpus))
        except Exception: pass
    except Exception:
        same = None
    if kind == "enc": ns, T = data.shape[0], data.shape[1]
    else: T, ns = data[1].shape
    r = {"value": round(one, 1), "unit": "frames/s", "cores": 1, "kind": "reference", "sample": "cpu_sample", "frames": int(nfr), "distinct_frames": int(ns * T), "same_work_value": same}
    if allc is not None: r["all_cores_value"] = round(allc, 1); r["all_cores"] = len(cpus)
    return r

def parity_sample(cfg, pcm, gpu_packets, gpu_lens, gpu_rng, gpu_dec=None):
    """After the timed region: the first CPU_STREAMS streams once more through the compiled reference the GPU path is bit-exact to (libopus_ref_fxa.so: fixed point + float
    API; libopus_ref_fx.so under --no-analysis) -- every frame in order, state carried -- and EVERY packet and final range of those streams compared with what the GPU
    batch produced during the run (warm-up and timed steps).  gpu_dec = (pcm [T, ns, frame*ch], final ranges [T, ns]): the decoder leg's output, compared with the
    reference decoder's on the same packets.  Returns {"ok", "streams", "frames", "mismatches"} or None when the checker did not travel."""
    path = os.path.join(ROOT, "oracle/_ref/libopus_ref_fx.so" if NO_ANALYSIS else "oracle/_ref/libopus_ref_fxa.so")
    if not os.path.exists(path): return None
    L = _ref_lib(path)
    L.opus_encoder_ctl.argtypes = None
    err = ctypes.c_int(); fr = frame_of(cfg)
    ns, T = pcm.shape[0], pcm.shape[1]
    out = (ctypes.c_ubyte * 1500)(); bad = 0; rv = ctypes.c_uint32()
    dpcm = np.zeros(fr * cfg["ch"], np.int16)
    for s in range(ns):
        st = L.opus_encoder_create(cfg["Fs"], cfg["ch"], cfg["app"], ctypes.byref(err))
        for req, v in cfg["ctls"]: L.opus_encoder_ctl(ctypes.c_void_p(st), ctypes.c_int(req), ctypes.c_int(v))
        dc = L.opus_decoder_create(cfg["Fs"], cfg["ch"], ctypes.byref(err)) if gpu_dec is not None else None
        for t in range(T):
            n = L.opus_encode(st, pcm[s, t].ctypes.data, fr, out, 1276)
            L.opus_encoder_ctl(ctypes.c_void_p(st), ctypes.c_int(4031), ctypes.byref(rv))
            ok = n == int(gpu_lens[t, s]) and rv.value == int(gpu_rng[t, s]) & 0xffffffff and bytes(out[:max(n, 0)]) == gpu_packets[t, s, :max(n, 0)].tobytes()
            if ok and dc is not None:
                m = L.opus_decode(dc, gpu_packets[t, s].ctypes.data, n, dpcm.ctypes.data, fr, 0)
                ok = m == fr and np.array_equal(dpcm, gpu_dec[0][t, s])
            bad += 0 if ok else 1
        L.opus_encoder_destroy(st)
        if dc is not None: L.opus_decoder_destroy(dc)
    return {"ok": bad == 0, "streams": ns, "frames": ns * T, "mismatches": bad, "checker": os.path.basename(path),
            "what": "packets + final ranges" + (" + decoded PCM" if gpu_dec is not None else "") + " of every frame of the sampled streams, GPU batch vs the compiled reference, after the timed region"}

def _ref_ms_encoder(L, NS, app):
    vp, ci = ctypes.c_void_p, ctypes.c_int
    L.opus_multistream_encoder_create.restype = vp; L.opus_multistream_encoder_create.argtypes = [ctypes.c_int32, ci, ci, ci, ctypes.c_char_p, ci, ctypes.POINTER(ci)]
    L.opus_multistream_encode.argtypes = [vp, vp, ci, vp, ctypes.c_int32]
    L.opus_multistream_encoder_destroy.argtypes = [vp]; L.opus_multistream_encoder_destroy.restype = None
    err = ci()
    st = L.opus_multistream_encoder_create(48000, NS, NS, 0, bytes(range(NS)), app, ctypes.byref(err))
    if not st or err.value: raise RuntimeError("opus_multistream_encoder_create (reference): %d" % err.value)
    L.opus_multistream_encoder_ctl.argtypes = [vp, ci, ci]
    for req, v in ((4002, NS * 64000), (4010, 10)): assert L.opus_multistream_encoder_ctl(st, req, v) == 0
    return st

def parity_sample_ms(cfg, pcm, gpu_packets, gpu_lens, gpu_rng, NS, max_bytes):
    """config 5 after the timed region: the first encoders of the batch once more through the compiled reference's opus_multistream_encode (255 mono streams, the same
    settings), every frame in order, every multistream packet and final range (the XOR over the streams) compared.  pcm [T][nb][frame][NS] int16."""
    path = os.path.join(ROOT, "oracle/_ref/libopus_ref_fx.so" if NO_ANALYSIS else "oracle/_ref/libopus_ref_fxa.so")
    if not os.path.exists(path): return None
    L = ctypes.CDLL(path, mode=ctypes.RTLD_LOCAL)
    T, nb = pcm.shape[0], pcm.shape[1]; fr = pcm.shape[2]
    out = (ctypes.c_ubyte * (max_bytes + 16))(); bad = 0; rv = ctypes.c_uint32()
    for b_ in range(nb):
        st = _ref_ms_encoder(L, NS, cfg["app"])
        for t in range(T):
            x = np.ascontiguousarray(pcm[t, b_])
            n = L.opus_multistream_encode(st, x.ctypes.data, fr, out, max_bytes)
            L.opus_multistream_encoder_ctl.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]; L.opus_multistream_encoder_ctl(st, 4031, ctypes.byref(rv)); L.opus_multistream_encoder_ctl.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
            ok = n == int(gpu_lens[t, b_]) and rv.value == int(gpu_rng[t, b_]) & 0xffffffff and bytes(out[:max(n, 0)]) == gpu_packets[t, b_, :max(n, 0)].tobytes()
            bad += 0 if ok else 1
        L.opus_multistream_encoder_destroy(st)
    return {"ok": bad == 0, "streams": nb * NS, "frames": nb * NS * T, "mismatches": bad, "checker": os.path.basename(path)}

def cpu_baseline_ms(cfg, pcm, NS, max_bytes, seconds=3.0):
    """one pinned core of the reference (float build) running opus_multistream_encode on the first encoder's frames, cycled (one encoder object, created outside the clock);
    elementary-stream frames/s"""
    path = os.path.join(ROOT, "oracle/_ref/libopus_ref_fl.so")
    if not os.path.exists(path): return None
    try: cpus = sorted(os.sched_getaffinity(0)); os.sched_setaffinity(0, {cpus[0]})
    except Exception: cpus = None
    L = ctypes.CDLL(path, mode=ctypes.RTLD_LOCAL)
    st = _ref_ms_encoder(L, NS, cfg["app"])
    T, fr = pcm.shape[0], pcm.shape[1]
    xs = [np.ascontiguousarray(pcm[t]) for t in range(T)]; out = (ctypes.c_ubyte * (max_bytes + 16))()
    n = 0; t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for x in xs: L.opus_multistream_encode(st, x.ctypes.data, fr, out, max_bytes)
        n += T
    dt = time.perf_counter() - t0
    L.opus_multistream_encoder_destroy(st)
    if cpus:
        try: os.sched_setaffinity(0, set(cpus))
        except Exception: pass
    return {"value": round(n * NS / dt, 1), "unit": "frames/s", "cores": 1, "kind": "reference", "sample": "cpu_sample_ms", "frames": int(n * NS), "distinct_frames": int(T * NS)}

def copy_bandwidth(dev):
    """device-to-device copy, GB/s of traffic (read + write)"""
    import torch
    n = 1 << 30
    a = torch.empty(n, dtype=torch.uint8, device=dev); b = torch.empty(n, dtype=torch.uint8, device=dev)
    b.copy_(a); torch.cuda.synchronize(dev)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record();
    for _ in range(4): b.copy_(a)
    e1.record(); torch.cuda.synchronize(dev)
    return 4 * 2 * n / (e0.elapsed_time(e1) * 1e-3) / 1e9

def run_config(cid, S, K, W, dev, local, rank, world, gather_cls=None, with_cpu=True, frames_per_launch=0, decode=False, gather_on=True):
    """times K frame-steps of BASELINE config `cid` on this rank -- the encoder, then (decode=True) the decoder on the packets the encoder just produced; returns a list of
    result dicts (rank-local figures; the caller reduces dt over ranks): [encode] or [encode, decode]"""
    import torch, torch.distributed as dist
    import opus_amd
    cfg = CONFIGS[cid]
    Fs, CH = cfg["Fs"], cfg["ch"]; FR = frame_of(cfg); TE = K + W
    T = max(TE, frames_per_launch if (world == 1 and cid == 2) else 0)                      # frames of input per stream (the T-frames-per-launch leg wants T of them)
    P = 256
    pool = synth(cfg, T, P, rank, corpus=CORPUS)
    P = pool.shape[0]
    pool_d = torch.from_numpy(pool).to(dev)
    g = torch.Generator(device="cpu"); g.manual_seed(1234 + rank)
    pid = torch.randint(0, P, (S,), generator=g).to(dev)
    off = (torch.randint(0, FR, (S,), generator=g) * CH).to(dev)
    gain = (0.5 + 0.5 * torch.rand((S,), generator=g)).to(dev)
    pcm = torch.empty((T, S, FR * CH), dtype=torch.int16, device=dev)
    ar = torch.arange(FR * CH, device=dev)
    for t in range(T):
        idx = off[:, None] + t * FR * CH + ar[None, :]
        x = pool_d[pid[:, None], idx].to(torch.float32) * gain[:, None]
        pcm[t] = x.round().clamp(-32768, 32767).to(torch.int16)
    del pool_d
    STRIDE = 1280 if cfg.get("frame_ms", 20) == 20 else 1344          # (a multi-frame call's slot: max_data_bytes + the staging head-room of the device-side assembly)
    NC = min(CPU_STREAMS, S)
    stream = torch.cuda.current_stream(dev)
    if cid == 5:
        # B multistream encoders x 255 mono AUDIO streams, resident on the device (opusgpu_ms_enc_batch_*, opus_amd/csrc/opus_ms_batch.h): channel split, the 65,535 elementary
        # encodes and the self-delimited packing are launches on one stream; `out` holds the B multistream packets
        NS = 255; Bn = S // NS; S = Bn * NS
        L = opus_amd.lib()
        L.opusgpu_ms_enc_batch_create.restype = ctypes.c_void_p
        L.opusgpu_ms_enc_batch_create.argtypes = [ctypes.c_int] * 6 + [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
        L.opusgpu_ms_enc_batch_ctl.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        L.opusgpu_ms_encode_batch_dev.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.opusgpu_ms_enc_batch_destroy.argtypes = [ctypes.c_void_p]; L.opusgpu_ms_enc_batch_destroy.restype = None
        err = ctypes.c_int()
        msb = L.opusgpu_ms_enc_batch_create(Bn, Fs, NS, 255, NS, 0, bytes(range(NS)), cfg["app"], local, ctypes.byref(err))
        if not msb: raise RuntimeError("opusgpu_ms_enc_batch_create failed: %d" % err.value)
        assert L.opusgpu_ms_enc_batch_ctl(msb, 4002, NS * 64000) == 0 and L.opusgpu_ms_enc_batch_ctl(msb, 4010, 10) == 0
        assert L.opusgpu_ms_enc_batch_ctl(msb, opus_amd.OPUS_AMD_SET_FLOAT_ANALYSIS_REQUEST, 0 if NO_ANALYSIS else 1) == 0
        pcm = pcm[:, :S].reshape(T, Bn, NS, FR).permute(0, 1, 3, 2).contiguous()            # [T][B][frame][channels] interleaved, as opus_multistream_encode takes it
        STRIDE = 65536; MS_MAX = (NS - 1) * 1279 + 7662 + 3 * NS + 8
        NP = Bn                                                                              # packets per step
        class _Ms:
            def encode_dev(self, p, fr, o, stride, l, r, hip_stream=None):
                rc = L.opusgpu_ms_encode_batch_dev(msb, p, fr, o, STRIDE, MS_MAX, l, r, hip_stream)
                if rc != 0: raise RuntimeError("opusgpu_ms_encode_batch_dev failed: %d" % rc)
            def close(self): L.opusgpu_ms_enc_batch_destroy(msb)
        b = _Ms()
    else:
        NP = S
        b = opus_amd.EncoderBatch(S, channels=CH, application=cfg["app"], Fs=Fs, device=local)
        for req, v in cfg["ctls"]: b.ctl(req, v)
        b.ctl(opus_amd.OPUS_AMD_SET_FLOAT_ANALYSIS_REQUEST, 0 if NO_ANALYSIS else 1)        # default: like the reference's default build (analysis.c + mlp.c at complexity 10)
        try: b.ctl(opus_amd.OPUS_AMD_SET_KERNEL_TIMING_REQUEST, 1)                         # HIP events between the launches of every call (a few event records per step): the last step's per-kernel times go into the line
        except Exception: pass
    # every step's packets stay on the device (the decoder leg and the parity sample read them): [TE][NP][STRIDE]
    pk = torch.zeros((TE, NP, STRIDE), dtype=torch.uint8, device=dev); lens = torch.zeros((TE, NP), dtype=torch.int32, device=dev); rng = torch.zeros((TE, NP), dtype=torch.int32, device=dev)
    # wire record of the final gather: its capacity per stream follows from the encoder settings (PacketGather.wire_capacity: the bitrate bound of these VBR streams; an overflow is reported)
    gather = None; gather_err = None; gather_fallback = None

    def step(t):
        b.encode_dev(pcm[t].data_ptr(), FR, pk[t].data_ptr(), STRIDE, lens[t].data_ptr(), rng[t].data_ptr(), hip_stream=stream.cuda_stream)

    for t in range(W): step(t)
    torch.cuda.synchronize(dev)
    if world > 1 and gather_cls and gather_on:
        # The exchange is warmed up on the warm-up steps' packets, one transport after the other ("auto": the RCCL gather first, then the point-to-point copies): a transport
        # that fails on ANY rank (all-reduced verdict: every rank takes the same decision) is replaced by the next one, so that the exchange north_star asks for is inside the
        # timed region whenever one of them works; the line says which one ran and why an earlier one did not.  Only when every transport fails is the exchange left out
        # (reported: "gather": {"error": ...}) rather than taking the run down.
        for tr in (["rccl", "p2p"] if GATHER == "auto" else [GATHER]):
            g = None; err = None
            try:
                if tr == "rccl" and os.environ.get("OPUS_AMD_BENCH_FAIL_RCCL") == "1": raise RuntimeError("forced failure of the RCCL gather (test hook OPUS_AMD_BENCH_FAIL_RCCL)")
                g = gather_cls(NP * world, STRIDE, dev, dst=0, bitrate_bps={2: 128000, 3: 24000, 4: 128000, 5: 255 * 64000}[cid], frame_rate=50, cbr=False, sub_streams=255 if cid == 5 else 1, transport=tr)
                for t in range(W): g.launch(lens[t], rng[t], pk[t])
                g.flush(); torch.cuda.synchronize(dev)
            except Exception as e: err = "%s: %s" % (type(e).__name__, e)
            okf = torch.tensor([0 if err else 1], dtype=torch.int32, device=dev if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(okf, op=dist.ReduceOp.MIN)
            if int(okf.item()) == 1: gather = g; gather_err = None; break
            err = err or "the exchange failed on another rank"
            gather_fallback = ((gather_fallback + "; ") if gather_fallback else "") + "%s: %s" % (tr, err)
            gather_err = gather_fallback
    if world > 1: dist.barrier()
    torch.cuda.synchronize(dev)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
    t0 = time.perf_counter()
    for k in range(K):
        ev[k][0].record(stream)
        step(W + k)
        ev[k][1].record(stream)
        if gather is not None: gather.launch(lens[W + k], rng[W + k], pk[W + k])  # the only exchange of the path: final gather of the packets (RCCL over xGMI), on a side stream, behind the next step's encode
    if gather is not None: gather.flush()                                          # ... and the last step's gather is inside the timed region too
    torch.cuda.synchronize(dev)
    if world > 1: dist.barrier()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    kern_ms = float(np.mean([e0.elapsed_time(e1) for e0, e1 in ev]))
    kernels_ms = {}
    if hasattr(b, "kernel_times"):
        try: kernels_ms = {k_: round(v_, 3) for k_, v_ in b.kernel_times().items()}           # the LAST timed step's launches, HIP events on the launch stream inside the library
        except Exception: kernels_ms = {}
    lens_h = lens[W:].cpu().numpy()
    ok = bool((lens_h > 0).all())
    mean_len = float(lens_h.mean()) / (255 if cid == 5 else 1)                               # config 5: per elementary stream (incl. its self-delimiting length byte)
    L = opus_amd.lib()
    L.opusgpu_enc_moved_state_bytes.restype = ctypes.c_int
    state_moved = L.opusgpu_enc_moved_state_bytes(cfg["app"], CH, 1 if cid == 4 else 0)          # state bytes read + written per frame-step
    L.opusgpu_enc_analysis_moved_bytes.restype = ctypes.c_int
    analysis_on = (not NO_ANALYSIS) and Fs >= 16000 and cfg["app"] != opus_amd.OPUS_APPLICATION_RESTRICTED_SILK and any(req == 4010 and v >= 10 for req, v in cfg["ctls"])
    if analysis_on: state_moved += L.opusgpu_enc_analysis_moved_bytes()                       # the tonality analysis' own state (phase history, 30 ms input window, band energies, info ring)
    alg = FR * CH * 2 + mean_len + 8 + state_moved
    res = {"config_id": cid, "leg": "encode", "workload": cfg["name"], "metric": cfg["metric"], "kernel": cfg["kernel"] + (" (+ oa_ms_split_kernel, oa_ms_pack_kernel)" if cid == 5 else ""), "streams_per_gpu": S, "dt": dt, "kernel_ms": kern_ms,
           "mean_packet_bytes": round(mean_len, 1), "all_packets_valid": ok, "algorithmic_bytes_per_frame": round(alg, 1), "state_bytes_moved_per_frame": state_moved, "float_analysis": bool(analysis_on),
           "kernels_ms": kernels_ms,
           "gather": ({"error": gather_err, "in_timed_region": False} if gather_err else None) if gather is None else dict(gather.stats(), in_timed_region=True, **({"fallback_from": gather_fallback} if gather_fallback else {}))}
    results = [res]
    # the CPU legs' sample: the first NC streams, every frame the batch saw of them
    pcm_h = None
    if with_cpu and cid != 5:
        pcm_h = np.ascontiguousarray(pcm[:TE, :NC].permute(1, 0, 2).cpu().numpy())          # [NC][TE][FR*CH]
        pk_h = pk[:, :NC].cpu().numpy(); lens_c = lens[:, :NC].cpu().numpy(); rng_c = rng[:, :NC].cpu().numpy()
        res["pcm_sample"] = pcm_h
        res["parity_sample"] = parity_sample(cfg, pcm_h, pk_h, lens_c, rng_c)
    if with_cpu and cid == 5:
        nb = min(2, Bn)
        ms_pcm = np.ascontiguousarray(pcm[:TE, :nb].cpu().numpy())                          # [TE][nb][FR][NS]
        res["parity_sample"] = parity_sample_ms(cfg, ms_pcm, pk[:, :nb].cpu().numpy(), lens[:, :nb].cpu().numpy(), rng[:, :nb].cpu().numpy(), NS, MS_MAX)
        res["ms_sample"] = (ms_pcm[:, 0], NS, MS_MAX)
    if frames_per_launch and world == 1 and cid == 2:
        # T consecutive frame-steps of every stream in ONE launch (the wave keeps its stream for T frames): SURVEY 8d "report also T = 50 consecutive steps"; a fresh batch
        Tn = min(frames_per_launch, T)
        b2 = opus_amd.EncoderBatch(S, channels=CH, application=cfg["app"], Fs=Fs, device=local)
        for req, v in cfg["ctls"]: b2.ctl(req, v)
        b2.ctl(opus_amd.OPUS_AMD_SET_FLOAT_ANALYSIS_REQUEST, 0 if NO_ANALYSIS else 1)
        outs = torch.zeros((Tn, S, STRIDE), dtype=torch.uint8, device=dev); lns = torch.zeros((Tn, S), dtype=torch.int32, device=dev); rgs = torch.zeros((Tn, S), dtype=torch.int32, device=dev)
        L.opusgpu_encode_batch_dev_frames.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        r = L.opusgpu_encode_batch_dev_frames(b2._b, pcm.data_ptr(), FR, Tn, outs.data_ptr(), STRIDE, 1276, lns.data_ptr(), rgs.data_ptr(), stream.cuda_stream)
        e1.record(stream); torch.cuda.synchronize(dev)
        if r == 0:
            m = min(Tn, TE); same = bool(torch.equal(lns[:m], lens[:m])) and bool(torch.equal(rgs[:m], rng[:m]))     # the T-frame launch reproduces the step-by-step run (lengths and final ranges of the frames both saw)
            res["frames_per_launch"] = {"T": Tn, "ms_per_frame_step": round(e0.elapsed_time(e1) / Tn, 3), "frames_per_s": round(S * Tn / (e0.elapsed_time(e1) * 1e-3), 1), "equals_step_by_step": same}
        b2.close(); del outs, lns, rgs
    if decode and cid != 5:
        # the decoder on these very packets: a decoder batch decodes them step by step with its state carried in HBM; timed like the encoder (HIP events per launch,
        # barrier + synchronize around the K steps)
        d = opus_amd.DecoderBatch(S, channels=CH, Fs=Fs, device=local)
        # the default call: oa_decode_look_kernel sorts the packets (one lane per stream), the CELT-only fast kernel and the general kernel each take their list -- whatever the
        # batch carries (configs 3 and 4 carry no CELT-only packet: their look finds the fast list empty).  OPUS_AMD_BENCH_DEC_NO_LOOK=1: opusgpu_dec_batch_set_fast_kernel(b, 0), A/B only
        fast_kernel = os.environ.get("OPUS_AMD_BENCH_DEC_NO_LOOK") != "1"
        if not fast_kernel: d.set_fast_kernel(False)
        dpcm = torch.zeros((TE, NC, FR * CH), dtype=torch.int16, device=dev)                 # (kept for the sampled streams only)
        work = torch.zeros((S, FR * CH), dtype=torch.int16, device=dev); dns = torch.zeros((S,), dtype=torch.int32, device=dev); drng = torch.zeros((TE, S), dtype=torch.int32, device=dev)
        def dstep(t):
            d.decode_dev(pk[t].data_ptr(), STRIDE, lens[t].data_ptr(), work.data_ptr(), FR, dns.data_ptr(), drng[t].data_ptr(), hip_stream=stream.cuda_stream)
        for t in range(W): dstep(t); dpcm[t].copy_(work[:NC])
        torch.cuda.synchronize(dev)
        if world > 1: dist.barrier()
        dev_ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
        t0 = time.perf_counter()
        for k in range(K):
            dev_ev[k][0].record(stream)
            dstep(W + k)
            dev_ev[k][1].record(stream)
            dpcm[W + k].copy_(work[:NC])                                                     # (64 streams' output kept for the parity sample: 245 KB per step)
        torch.cuda.synchronize(dev)
        if world > 1: dist.barrier()
        torch.cuda.synchronize(dev)
        r2 = dict(res); r2.pop("pcm_sample", None); r2.pop("frames_per_launch", None); r2.pop("kernels_ms", None)
        r2["leg"] = "decode"; r2["dt"] = time.perf_counter() - t0
        r2["kernel_ms"] = float(np.mean([a.elapsed_time(b_) for a, b_ in dev_ev]))
        r2["kernel"] = "oa_decode_look_kernel + oa_sdec_lane_kernel (SILK steady state, lane = stream) + oa_decode_fast_kernel / oa_decode_hyb_kernel (CELT frames up to their bands) + oa_celt_dpvq_kernel (the bands, four streams per wave) + oa_celt_dback_kernel (synthesis) + oa_decode_kernel (one call)" if fast_kernel else "oa_decode_kernel (opusgpu_dec_batch_set_fast_kernel(b, 0))"
        r2["dec_fast_kernel"] = bool(fast_kernel)
        r2["all_packets_valid"] = bool((dns.cpu().numpy() == FR).all()) and bool(torch.equal(drng, rng))      # every stream decoded FR samples and every frame ends on the encoder's final range
        L.opusgpu_dec_state_size.restype = ctypes.c_int
        r2["algorithmic_bytes_per_frame"] = round(FR * CH * 2 + mean_len + 8 + 2 * L.opusgpu_dec_state_size(), 1)
        r2["state_bytes_moved_per_frame"] = 2 * L.opusgpu_dec_state_size()
        r2["metric"] = "decoded frames/s (" + cfg["metric"].split("(", 1)[1]
        if pcm_h is not None:
            r2["parity_sample"] = parity_sample(cfg, pcm_h, pk_h, lens_c, rng_c, gpu_dec=(dpcm.cpu().numpy(), None))
            r2["dec_sample"] = (pk_h, lens_c)
        d.close(); del dpcm, work
        results.append(r2)
    b.close()
    del pcm, pk
    torch.cuda.empty_cache()
    return results

def steady_state_leg(cid, dev, local, K, S_full=65536, S1=1024, T1=500, with_cpu=True):
    """Steady state next to the cold-start figure: S1 streams live through T1 CONSECUTIVE frames (10 s of audio each, state on the device) -- VBR reservoir,
    prefilter and energy memories, the analysis' history all filled -- timed per step at the head and at the tail of the run; four of the streams checked against the compiled
    reference over all T1 frames; then the S1 warmed-up states are fanned out over S_full streams (opusgpu_enc_batch_copy_states, 64 replicas each) and K more steps are timed
    at full width on the frames that follow: the throughput of a batch that has been running for ten seconds."""
    import torch, opus_amd
    cfg = CONFIGS[cid]; Fs, CH = cfg["Fs"], cfg["ch"]; FR = Fs // 50; P = 32; STRIDE = 1280
    T = T1 + K
    pool = torch.from_numpy(synth(cfg, T, P, 4242, corpus=CORPUS)).to(dev)                  # [P][(T+2)*FR*CH]: P tunes entered at evenly spread points of the piece
    g = torch.Generator(device="cpu"); g.manual_seed(99)
    pid = torch.randint(0, P, (S1,), generator=g).to(dev); off = (torch.randint(0, FR, (S1,), generator=g) * CH).to(dev); gain = (0.5 + 0.5 * torch.rand((S1,), generator=g)).to(dev)
    ar = torch.arange(FR * CH, device=dev)
    def frames(t):
        x = pool[pid[:, None], off[:, None] + t * FR * CH + ar[None, :]].to(torch.float32) * gain[:, None]
        return x.round().clamp(-32768, 32767).to(torch.int16)
    stream = torch.cuda.current_stream(dev)
    def make(S):
        b = opus_amd.EncoderBatch(S, channels=CH, application=cfg["app"], Fs=Fs, device=local)
        for req, v in cfg["ctls"]: b.ctl(req, v)
        b.ctl(opus_amd.OPUS_AMD_SET_FLOAT_ANALYSIS_REQUEST, 0 if NO_ANALYSIS else 1)
        return b
    b1 = make(S1)
    pk = torch.zeros((T1, S1, STRIDE), dtype=torch.uint8, device=dev); lens = torch.zeros((T1, S1), dtype=torch.int32, device=dev); rng = torch.zeros((T1, S1), dtype=torch.int32, device=dev)
    keep = torch.zeros((T1, 4, FR * CH), dtype=torch.int16, device=dev)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(T1)]
    for t in range(T1):
        x = frames(t); keep[t] = x[:4]
        ev[t][0].record(stream)
        b1.encode_dev(x.data_ptr(), FR, pk[t].data_ptr(), STRIDE, lens[t].data_ptr(), rng[t].data_ptr(), hip_stream=stream.cuda_stream)
        ev[t][1].record(stream)
    torch.cuda.synchronize(dev)
    ms = np.array([a.elapsed_time(b_) for a, b_ in ev]); ln = lens.cpu().numpy()
    out = {"streams": S1, "consecutive_frames": T1, "ms_per_step_first_25": round(float(ms[:25].mean()), 3), "ms_per_step_last_25": round(float(ms[-25:].mean()), 3),
           "mean_packet_bytes_first_25": round(float(ln[:25].mean()), 1), "mean_packet_bytes_last_25": round(float(ln[-25:].mean()), 1), "all_packets_valid": bool((ln > 0).all())}
    if with_cpu:
        ps = parity_sample(cfg, np.ascontiguousarray(keep.permute(1, 0, 2).cpu().numpy()), pk[:, :4].cpu().numpy(), ln[:, :4], rng[:, :4].cpu().numpy())
        out["parity_sample_ok"] = None if ps is None else ps["ok"]; out["parity_frames"] = None if ps is None else ps["frames"]
    # fan the warmed-up states out and time K steps at full width on the frames that follow
    R = S_full // S1
    b2 = make(S_full)
    for r in range(R): b2.copy_states_from(b1, S1, dst_first=r * S1)
    pk2 = torch.zeros((S_full, STRIDE), dtype=torch.uint8, device=dev); l2 = torch.zeros((K, S_full), dtype=torch.int32, device=dev); r2 = torch.zeros((S_full,), dtype=torch.int32, device=dev)
    xs = [frames(T1 + k).repeat(R, 1) for k in range(K)]
    torch.cuda.synchronize(dev)
    ev2 = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
    t0 = time.perf_counter()
    for k in range(K):
        ev2[k][0].record(stream)
        b2.encode_dev(xs[k].data_ptr(), FR, pk2.data_ptr(), STRIDE, l2[k].data_ptr(), r2.data_ptr(), hip_stream=stream.cuda_stream)
        ev2[k][1].record(stream)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    l2h = l2.cpu().numpy()
    out["full_width"] = {"streams": S_full, "steps": K, "value": round(S_full * K / dt, 1), "unit": "frames/s", "ms_per_step": round(dt / K * 1e3, 3),
                         "kernel_ms": round(float(np.mean([a.elapsed_time(b_) for a, b_ in ev2])), 3), "mean_packet_bytes": round(float(l2h.mean()), 1), "all_packets_valid": bool((l2h > 0).all()),
                         "replicas_agree": bool((l2h.reshape(K, R, S1) == l2h.reshape(K, R, S1)[:, :1]).all())}
    b1.close(); b2.close()
    del pool, pk, pk2, xs
    torch.cuda.empty_cache()
    return out

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--streams", type=int, default=0, help="streams per GPU (default 65,536; config 5: 257 x 255)")
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS), help="BASELINE.json configuration (2 = headline)")
    ap.add_argument("--frames-per-launch", type=int, default=-1, help="also time T consecutive frame-steps in one launch (config 2, N = 1; default 50 in the default run, 0 = off)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-analysis", action="store_true", help="encode like a reference built with DISABLE_FLOAT_API: no tonality / music analysis at complexity 10 (the round-1/2 workload)")
    ap.add_argument("--corpus", default="reference", choices=["pool", "reference"], help="input signals of the 48 kHz configurations: the reference's generate_music() tunes (tests/test_opus_encode.c:57; default) or this repo's music / noise-burst pool")
    ap.add_argument("--decode", action="store_true", help="time the decoder on the packets of the chosen configuration (encoded first) and report IT as the main line")
    ap.add_argument("--no-extra-configs", action="store_true", help="N = 1 default run: skip the config 3 / 4 / 5 and decoder legs")
    ap.add_argument("--steady-state", type=int, default=-1, help="also run the steady-state leg of the main configuration: 1,024 streams through this many consecutive frames, then a full-width batch of the warmed-up states (default 500 in the default run, 0 = off)")
    ap.add_argument("--no-gather", action="store_true", help="N > 1: leave the final packet gather out (to time what it costs)")
    ap.add_argument("--gather", default="auto", choices=["auto", "rccl", "p2p"], help="N > 1: auto (default) = the RCCL gather, and when its warm-up fails on any rank the point-to-point copies instead, so that the exchange is in the timed region either way; rccl = one RCCL gather collective per step; p2p = every rank copying its record into rank 0's buffer through an IPC mapping (no collective; independent of ProcessGroupNCCL)")
    a = ap.parse_args()
    global CORPUS, NO_ANALYSIS, GATHER
    CORPUS = a.corpus; NO_ANALYSIS = a.no_analysis; GATHER = a.gather
    import torch, torch.distributed as dist
    import opus_amd
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and world == 1:
        print("bench.py: launch with torch.distributed.run for --gpus > 1", file=sys.stderr); sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible — the product path has no CPU fallback", file=sys.stderr); sys.exit(3)
    # test hook (tests/test_gpu_bench_ranks.py): OPUS_AMD_BENCH_BACKEND=gloo runs the N > 1 code path with every rank on GPU 0 and the exchange staged through host
    # memory, so that the sharded bench is executed end to end on a 1-GPU box; the driver's multi-GPU runs use RCCL ("nccl"), one GPU per rank
    backend = os.environ.get("OPUS_AMD_BENCH_BACKEND", "nccl")
    if backend != "nccl": local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl": dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else: dist.init_process_group(backend, rank=rank, world_size=world)
    from opus_amd.shard import PacketGather
    S = a.streams or (257 * 255 if a.config == 5 else 65536)
    K, W = a.steps, a.warmup
    full = world == 1 and a.config == 2 and not a.no_extra_configs and not a.decode       # the default N = 1 run: every BASELINE configuration + the decoder, one JSON line
    fpl = a.frames_per_launch if a.frames_per_launch >= 0 else (50 if full else 0)
    cpu_on = rank == 0 and world == 1 and not a.no_cpu_baseline
    legs = run_config(a.config, S, K, W, dev, local, rank, world, gather_cls=PacketGather, with_cpu=cpu_on, frames_per_launch=fpl, decode=a.decode or full, gather_on=not a.no_gather)
    main_res = legs[-1] if a.decode else legs[0]
    tt = torch.tensor([main_res["dt"]], dtype=torch.float64, device=dev)
    per_rank = None
    if world > 1:
        if backend != "nccl": tt = tt.cpu()
        mine = torch.tensor([float(rank), float(local), main_res["dt"] / K * 1e3, main_res["kernel_ms"]], dtype=torch.float64, device=tt.device)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [{"rank": int(x[0].item()), "device": int(x[1].item()), "ms_per_step": round(float(x[2].item()), 3), "kernel_ms": round(float(x[3].item()), 3)} for x in allr]
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
    extras = [l for l in legs if l is not main_res]
    if full:
        Kx = max(3, K // 2)
        for cid in (3, 4):
            extras += run_config(cid, S, Kx, 2, dev, local, rank, world, with_cpu=cpu_on, decode=True)
        for cid in (31, 32):
            extras += run_config(cid, S, Kx, 2, dev, local, rank, world, with_cpu=cpu_on)
        extras += run_config(5, (a.streams // 255) * 255 if a.streams else 257 * 255, Kx, 2, dev, local, rank, world, with_cpu=cpu_on)
    steady = None
    ss_frames = a.steady_state if a.steady_state >= 0 else (500 if full else 0)
    if ss_frames and world == 1 and a.config != 5 and not a.decode:
        steady = steady_state_leg(a.config, dev, local, K, S_full=S if S % 1024 == 0 and S >= 1024 else 65536, T1=ss_frames, with_cpu=cpu_on)
    if rank == 0:
        peak_meas = round(copy_bandwidth(dev), 1) if world == 1 else None
        built = opus_amd.lib().opusgpu_build_info().decode()
        try: src_now = opus_amd.source_hash()
        except Exception: src_now = None
        traffic_docs = []
        for name in (["pmc_traffic_r02.json"] if NO_ANALYSIS else ["pmc_traffic_r06.json", "pmc_traffic_r05.json", "pmc_traffic_r04.json", "pmc_traffic_r03.json"]):
            try: traffic_docs.append((name, json.load(open(os.path.join(ROOT, "profiles", name)))))
            except Exception: continue
        used_traffic = set()
        def roof(r, Sn, short=False):
            ach = Sn * r["algorithmic_bytes_per_frame"] / (r["kernel_ms"] * 1e-3) / 1e9
            traffic = None; issue = None
            key = ("decode_%d" if r["leg"] == "decode" else "config_%d") % r["config_id"]
            for name, doc in traffic_docs:
                try:
                    pt = doc[key]
                    if pt.get("hbm_bytes_per_frame"): traffic = int(pt["hbm_bytes_per_frame"] * Sn)
                    issue = {"valu_busy": pt["issue"].get("valu_busy_per_simd"), "lanes": pt["lane_utilisation"]["active_lanes_per_valu_cycle"]}
                    used_traffic.add(name)
                    break
                except Exception: continue
            o = {"bound": "hbm", "achieved": round(ach, 2), "peak": 8000.0, "unit": "GB/s", "frac": round(ach / 8000.0, 5), "traffic": traffic, "kernel_ms": round(r["kernel_ms"], 3),
                 "algorithmic_bytes_per_frame": r["algorithmic_bytes_per_frame"]}
            if issue: o["issue"] = issue
            km = r.get("kernels_ms") or {}
            if km:
                dk = max(km, key=km.get)
                if not short: o["kernels_ms"] = km
                # (the extra legs carry the dominant kernel only: the driver keeps the last 8 KB of the line; profiles/r06_final has every kernel of every leg)
                o["dominant"] = {"kernel": dk, "ms": km[dk], "achieved": round(Sn * r["algorithmic_bytes_per_frame"] / (km[dk] * 1e-3) / 1e9, 2), "frac": round(Sn * r["algorithmic_bytes_per_frame"] / (km[dk] * 1e-3) / 1e9 / 8000.0, 5)}
            return o
        def cpu_leg(r, seconds, allc):
            if r["leg"] == "decode" and "dec_sample" in r: return cpu_baseline(CONFIGS[r["config_id"]], r["dec_sample"], seconds=seconds, all_cores_seconds=allc, kind="dec")
            if r["leg"] == "encode" and "pcm_sample" in r: return cpu_baseline(CONFIGS[r["config_id"]], r["pcm_sample"], seconds=seconds, all_cores_seconds=allc, kind="enc")
            if r["leg"] == "encode" and "ms_sample" in r: return cpu_baseline_ms(CONFIGS[r["config_id"]], *r["ms_sample"], seconds=seconds)
            return None
        frames = main_res["streams_per_gpu"] * world * K
        model, ncpu = host_info()
        res = {
            "metric": main_res["metric"] if (a.decode or a.config != 2) else "encoded frames/s (48 kHz stereo, 20 ms, complexity 10)", "value": round(frames / dt, 1), "unit": "frames/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(dt / K * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "i32", "data": ("synthetic: the reference's generate_music() (tests/test_opus_encode.c:57), 256 seeds entering the 30 s piece at spread points" if (a.corpus == "reference" and CONFIGS[a.config]["Fs"] == 48000) else "synthetic"),
            "config": {"workload": ("DECODE of the packets of: " if a.decode else "") + CONFIGS[a.config]["name"] + ", bit-exact fixed-point" + (" + float-API tonality analysis (the reference's default build)" if main_res.get("float_analysis") else " (DISABLE_FLOAT_API)"), "baseline_config": a.config, "float_analysis": bool(main_res.get("float_analysis")),
                       "streams_per_gpu": main_res["streams_per_gpu"], "frames_per_step": main_res["streams_per_gpu"] * world, "mean_packet_bytes": main_res["mean_packet_bytes"], "all_packets_valid": main_res["all_packets_valid"],
                       "parity_sample_ok": None if not main_res.get("parity_sample") else main_res["parity_sample"]["ok"], "parity_sample": main_res.get("parity_sample"),
                       "lib_build": built, "lib_matches_sources": None if src_now is None else built == "OA_SRC_HASH=" + src_now,
                       "parallelism": "streams sharded over %d GPU(s), no data-path collective%s" % (world, (", final gather FAILED in the warm-up and was left out (see \"gather\")" if (main_res.get("gather") or {}).get("error") else (", final gather of the compacted packets in the timed region (%s; side stream, double-buffered)" % ((main_res.get("gather") or {}).get("transport") or a.gather)) if not a.no_gather else ", final gather switched off (--no-gather)") if world > 1 else "")},
            "roofline": (lambda ro_: dict(ro_, kernel=(ro_["dominant"]["kernel"] + " (dominant, see roofline.dominant; achieved / frac are the whole call's, all of kernels_ms)") if ro_.get("dominant") else main_res["kernel"], peak_measured=peak_meas))(roof(main_res, main_res["streams_per_gpu"])),
        }
        if per_rank is not None: res["ranks_seen"] = len(per_rank); res["per_rank"] = per_rank
        if main_res.get("gather"): res["gather"] = main_res["gather"]
        if "frames_per_launch" in legs[0]: res["frames_per_launch"] = legs[0]["frames_per_launch"]
        if steady: res["steady_state"] = steady
        c = cpu_leg(main_res, 10.0, 4.0) if cpu_on else None
        if c:
            c["host_nproc"] = ncpu; c["cpu_model"] = model
            res["cpu_baseline"] = c
            if c["value"]: res["speedup_vs_cpu_1core"] = round(res["value"] / c["value"], 2)
        if extras:
            res["configs"] = {}
            for r in extras:
                Kx = K if r["config_id"] == a.config else max(3, K // 2)
                e = {"value": round(r["streams_per_gpu"] * Kx / r["dt"], 1), "ms_per_step": round(r["dt"] / Kx * 1e3, 3), "steps": Kx,
                     "streams": r["streams_per_gpu"], "mean_packet_bytes": r["mean_packet_bytes"], "valid": r["all_packets_valid"], "parity_ok": None if not r.get("parity_sample") else r["parity_sample"]["ok"],
                     "roofline": {k_: v_ for k_, v_ in roof(r, r["streams_per_gpu"], short=True).items() if k_ not in ("bound", "peak", "unit")}}
                if "dec_fast_kernel" in r and not r["dec_fast_kernel"]: e["dec_fast_kernel"] = False
                c = cpu_leg(r, 3.0, 0) if cpu_on else None
                if c:
                    e["cpu"] = {k_: c[k_] for k_ in ("value", "same_work_value", "frames") if k_ in c}
                    if c["value"]: e["x_cpu_1core"] = round(e["value"] / c["value"], 1)
                if CONFIGS[r["config_id"]].get("frame_ms", 20) != 20: e["frames_20ms_per_s"] = round(e["value"] * CONFIGS[r["config_id"]]["frame_ms"] / 20, 1)
                res["configs"][CONFIGS[r["config_id"]].get("key") or (("decode_%d" if r["leg"] == "decode" else "config_%d") % r["config_id"])] = e
        # what every leg has in common, said once (the per-leg entries stay short: the driver keeps the last 8 KB of this line)
        res["notes"] = {
            "legs": "configs.<config_N|decode_N>: BASELINE.json configuration N at 20 ms, complexity 10, 65,536 streams (decode_N: the decoder on its packets; config_5: elementary-stream frames/s; config_3_fec / config_3_60ms: config 3 with in-band FEC at 10 % loss / in 60 ms packets, calls/s); DESIGN.md 5 has the definitions",
            "roofline": "achieved = algorithmic bytes per frame x streams / kernel_ms (HIP events on the launch stream around one call, in the timed region), frac = / 8000 GB/s; kernels_ms / dominant: HIP events between the call's launches inside the library (last timed step); traffic / issue: rocprofv3 counter passes, profiles/%s" % (", ".join(sorted(used_traffic)) or "none on this box"),
            "cpu": "cpu_baseline / cpu: libopus float + RTCD on ONE pinned core, one codec state through >= %d consecutive frames of streams 0..%d (config_5: opus_multistream_encode on encoder 0); same_work_value = the fixed-point build the GPU path is bit-exact to" % (CPU_MIN_FRAMES, CPU_STREAMS - 1),
            "parity": "parity_ok / parity_sample: after the timed region streams 0..%d (config_5: 510 streams) once more through the compiled reference: every packet and final range (decode legs: + PCM) equal" % (CPU_STREAMS - 1),
            "steady_state": "1,024 streams x 500 consecutive frames on the device (4 x 500 checked against the reference), then fanned out over 65,536 streams and timed",
        }
        print(json.dumps(res))
    if world > 1: dist.destroy_process_group()

if __name__ == "__main__":
    main()
