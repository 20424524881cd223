"""CPU: the encoder WITH its tonality / music analysis (the library's default, complexity 10) against the compiled reference built the way a fixed-point libopus is by
default -- FIXED_POINT with the float API, oracle/_ref/libopus_ref_fxa.so (src/analysis.c + src/mlp.c active) -- through the classic API on the wave emulator:
packet bytes and final range call by call.  What the analysis steers is what is exercised: the allocation tuning of CELT-coded frames (config 2), the mode and
bandwidth decisions of unforced AUDIO / VOIP encoders, the generalised DTX, calls of 2.5 ... 120 ms, the 24-bit and float entry points (which hand the analysis
un-rounded samples), controls changed mid-stream (complexity in and out of 10: the analysis state is dropped, src/opus_encoder.c:1262).
tests/test_gpu_analysis.py re-runs it on the MI355X against opus_amd/libopus_amd.so."""
import ctypes, numpy as np, pytest
import capi, signals
from reflib import ref_fx, ref_fxa
from test_hostemu_encoder_modes import sig_for
pytestmark = pytest.mark.skipif(ref_fx() is None or ref_fxa() is None, reason="oracle/_ref not built")
WHICH = "emu"
FLOAT_ANALYSIS = 11900

def pair(Fs, ch, app, **ctl):
    ctl.setdefault("complexity", 10)
    a = capi.Enc("ref_fxa", Fs, ch, app, **ctl); b = capi.Enc(WHICH, Fs, ch, app, **ctl)
    b.L.opus_encoder_ctl.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    assert b.L.opus_encoder_ctl(b.st, FLOAT_ANALYSIS, 1) == 0
    return a, b

def run(Fs, ch, app, frames, schedule=None, seed=0, maxb=1276, sig=None, **ctl):
    a, b = pair(Fs, ch, app, **ctl)
    if sig is None: sig = sig_for(Fs, ch, sum(frames) + 16, seed)
    pos = 0; modes = []
    for i, fr in enumerate(frames):
        if schedule and i in schedule:
            for k, v in schedule[i].items():
                if k == "reset":
                    for e in (a, b): e.L.opus_encoder_ctl.argtypes = [ctypes.c_void_p, ctypes.c_int]; assert e.L.opus_encoder_ctl(e.st, 4028) == 0
                else: ra = a.set(k, v); rb = b.set(k, v); assert ra == rb, (i, k, v, ra, rb)
        pcm = sig[pos:pos + fr]; pos += fr
        x = a.encode(pcm, fr, maxb); y = b.encode(pcm, fr, maxb)
        modes.append("?" if x[1] <= 0 else "d" if x[1] <= 2 else "C" if x[0][0] & 0x80 else "H" if (x[0][0] & 0x60) == 0x60 else "S")
        assert x[1] == y[1], (i, fr, "".join(modes), x[1], y[1])
        assert x[2] == y[2], (i, fr, "".join(modes), hex(x[2]), hex(y[2]))
        assert x[0] == y[0], (i, fr, "".join(modes), [k for k in range(len(x[0])) if x[0][k] != y[0][k]][:8])
    return "".join(modes)

def music(Fs, ch, nsamp, seed):
    s = signals.music(nsamp * (48000 // Fs) // 960 + 2, seed=seed)[::48000 // Fs][:nsamp]
    return np.ascontiguousarray(s if ch == 2 else s[:, :1])

def test_the_analysis_changes_config_2():
    """the reason this suite exists: with the analysis the packets of config 2 differ from the DISABLE_FLOAT_API build's in (nearly) every frame"""
    Fs, ch = 48000, 2
    s = music(Fs, ch, 960 * 40, 3)
    a = capi.Enc("ref_fxa", Fs, ch, 2051, bitrate=128000, complexity=10); c = capi.Enc("ref", Fs, ch, 2051, bitrate=128000, complexity=10)
    differ = sum(a.encode(s[i * 960:(i + 1) * 960], 960)[0] != c.encode(s[i * 960:(i + 1) * 960], 960)[0] for i in range(40))
    assert differ > 30

@pytest.mark.parametrize("ch", [1, 2])
def test_config_2_with_analysis(ch):
    run(48000, ch, 2051, [960] * 70, sig=music(48000, ch, 960 * 71, 3 + ch), bitrate=64000 * ch)
    run(48000, ch, 2051, [960] * 40, seed=5, bitrate=48000 * ch)

def test_lowdelay_rates_and_sizes():
    run(48000, 2, 2051, [480] * 40 + [240] * 20 + [120] * 20 + [960] * 10, sig=music(48000, 2, 960 * 60, 7), bitrate=96000)
    run(24000, 2, 2051, [480] * 50, sig=music(24000, 2, 480 * 52, 8), bitrate=64000)
    run(16000, 1, 2051, [320] * 50, seed=9, bitrate=32000)
    run(48000, 2, 2051, [1920, 2880, 960, 3840, 4800, 5760, 1920, 960, 2880], sig=music(48000, 2, 960 * 30, 10), bitrate=128000)
    run(12000, 1, 2051, [240] * 10, seed=11, bitrate=24000)                     # below 16 kHz the analysis does not run (:1249)

def test_audio_unforced_mode_decisions():
    m = run(48000, 2, 2049, [960] * 120, sig=music(48000, 2, 960 * 121, 12), bitrate=64000)
    assert "C" in m
    m = run(48000, 1, 2049, [960] * 120, seed=13, bitrate=24000)                 # speech at a rate where the music probability decides SILK / hybrid / CELT
    m = run(48000, 2, 2049, [960] * 60 + [1920] * 10 + [2880] * 6, seed=14, bitrate=40000)
    m = run(16000, 1, 2048, [320] * 80, seed=15, bitrate=20000)
    m = run(24000, 1, 2049, [480] * 80, sig=music(24000, 1, 480 * 81, 16), bitrate=32000)

def test_speech_music_speech():
    Fs, ch = 48000, 1
    sp = sig_for(Fs, ch, 960 * 50, 17); mu = music(Fs, ch, 960 * 60, 18)
    s = np.concatenate([sp, mu, sp])
    m = run(Fs, ch, 2049, [960] * 155, sig=s, bitrate=32000)
    assert len(set(m)) >= 2, m

def test_forced_modes_and_hybrid():
    run(48000, 2, 2049, [960] * 40, seed=19, bitrate=128000, force_mode=1001, bandwidth=1105)
    run(48000, 1, 2049, [960] * 30, {10: dict(force_mode=1002), 20: dict(force_mode=1000)}, seed=20, bitrate=48000, force_mode=1001)
    run(16000, 1, 2048, [320] * 30, seed=21, bitrate=24000, force_mode=1000, bandwidth=1103)

def test_dtx_with_activity_probability():
    Fs, ch = 48000, 1
    s = sig_for(Fs, ch, 960 * 140, 22).copy()
    s[960 * 30:960 * 70] = (s[960 * 30:960 * 70].astype(np.int32) // 512).astype(np.int16)      # near-silence: the activity probability drops, not digital silence
    s[960 * 90:960 * 110] = 0
    m = run(Fs, ch, 2049, [960] * 138, sig=s, bitrate=32000, dtx=1)
    m2 = run(Fs, ch, 2051, [960] * 138, sig=s, bitrate=48000, dtx=1)
    m3 = run(16000, ch, 2048, [320] * 138, sig=np.ascontiguousarray(s[::3]), bitrate=16000, dtx=1)
    assert "d" in m + m2 + m3

def test_cbr_and_constrained():
    run(48000, 2, 2051, [960] * 30, sig=music(48000, 2, 960 * 31, 23), bitrate=96000, vbr=0)
    run(48000, 2, 2049, [960] * 30, sig=music(48000, 2, 960 * 31, 24), bitrate=96000, vbr=1, vbr_constraint=0)
    run(48000, 1, 2049, [960] * 30, seed=25, bitrate=20000, vbr=0)

def test_controls_midstream():
    sched = {10: dict(complexity=9), 20: dict(complexity=10), 30: dict(signal=3001), 40: dict(signal=-1000), 50: dict(reset=1), 60: dict(bandwidth=1103), 70: dict(bandwidth=-1000),
             80: dict(signal=3002), 90: dict(max_bandwidth=1104, bitrate=24000)}
    run(48000, 2, 2049, [960] * 100, sched, sig=music(48000, 2, 960 * 101, 26), bitrate=64000)
    run(48000, 2, 2051, [960] * 100, sched, seed=27, bitrate=64000)

def test_signal_type_steers_the_lowdelay_application():
    """OPUS_SET_SIGNAL moves the stereo -> mono and bandwidth thresholds of RESTRICTED_LOWDELAY too (voice_est, src/opus_encoder.c:1413), analysis or not"""
    for sig in (3001, 3002):
        run(48000, 2, 2051, [960] * 12, seed=28, bitrate=18000, signal=sig, complexity=5)
        run(48000, 2, 2051, [960] * 12, seed=28, bitrate=11000, signal=sig)

def _encode_any(e, fn, arr, fr, ctype_ptr, maxb=1276):
    out = (ctypes.c_ubyte * 1500)()
    getattr(e.L, fn).argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    n = getattr(e.L, fn)(e.st, arr.ctypes.data, fr, out, maxb)
    return n, bytes(out[:max(n, 0)]), e.get(4031) & 0xffffffff

def test_24_bit_and_float_entry_points_feed_the_analysis_unrounded_samples():
    Fs, ch = 48000, 2
    base = music(Fs, ch, 960 * 50, 29).astype(np.float64)
    rng = np.random.default_rng(30)
    fine = base + rng.uniform(-0.5, 0.5, base.shape)                                       # sub-LSB detail that the int16 path never sees
    for fn, arr in (("opus_encode24", np.round(fine * 256).astype(np.int32)), ("opus_encode_float", (fine / 32768.0).astype(np.float32))):
        for app, rate in ((2051, 96000), (2049, 48000)):
            a, b = pair(Fs, ch, app, bitrate=rate)
            for i in range(45):
                x = _encode_any(a, fn, np.ascontiguousarray(arr[i * 960:(i + 1) * 960]), 960, None); y = _encode_any(b, fn, np.ascontiguousarray(arr[i * 960:(i + 1) * 960]), 960, None)
                assert x == y, (fn, app, i, x[0], y[0])

def test_multistream_with_analysis():
    """every elementary encoder of a multistream encoder runs its own analysis on its own channels (opus_multistream_encoder.c:1027, c1 / c2 of the stream)"""
    R, E = capi.load("ref_fxa"), capi.load(WHICH)
    vp, ci = ctypes.c_void_p, ctypes.c_int
    Fs, nch = 48000, 3
    sig = np.concatenate([music(Fs, 2, 960 * 40, 31), sig_for(Fs, 1, 960 * 40, 32)], axis=1).astype(np.int16)
    for app, rate in ((2049, 96000), (2051, 160000)):
        out = []
        for L in (R, E):
            L.opus_multistream_encoder_create.restype = vp
            L.opus_multistream_encoder_create.argtypes = [ci, ci, ci, ci, ctypes.c_char_p, ci, ctypes.POINTER(ci)]
            L.opus_multistream_encode.argtypes = [vp, vp, ci, vp, ci]
            L.opus_multistream_encoder_destroy.argtypes = [vp]; L.opus_multistream_encoder_destroy.restype = None
            L.opus_multistream_encoder_ctl.argtypes = [vp, ci, ci]
            err = ci()
            e = L.opus_multistream_encoder_create(Fs, nch, 2, 1, bytes([0, 1, 2]), app, ctypes.byref(err)); assert e and err.value == 0
            assert L.opus_multistream_encoder_ctl(e, 4010, 10) == 0 and L.opus_multistream_encoder_ctl(e, 4002, rate) == 0
            if L is E: assert L.opus_multistream_encoder_ctl(e, FLOAT_ANALYSIS, 1) == 0
            buf = (ctypes.c_ubyte * 4000)(); seq = []
            for i in range(38):
                p = np.ascontiguousarray(sig[i * 960:(i + 1) * 960])
                n = L.opus_multistream_encode(e, p.ctypes.data, 960, buf, 4000); seq.append(bytes(buf[:max(n, 0)]) if n > 0 else n)
            L.opus_multistream_encoder_destroy(e); out.append(seq)
        bad = [i for i in range(38) if out[0][i] != out[1][i]]
        assert not bad, (app, bad[:5])

def test_multistream_float_and_projection_with_analysis():
    """the float entry point of the multistream encoder hands every elementary analysis its channels un-rounded; a projection (ambisonics) encoder's analyses look at the
    caller's un-mixed channels (opus_multistream_encoder.c:1027 passes the original pcm with the stream's channel indices)"""
    R, E = capi.load("ref_fxa"), capi.load(WHICH)
    vp, ci = ctypes.c_void_p, ctypes.c_int
    Fs = 48000
    rng = np.random.default_rng(41)
    base = np.concatenate([music(Fs, 2, 960 * 30, 40), sig_for(Fs, 2, 960 * 30, 42)], axis=1).astype(np.float64)          # 4 channels
    fine = ((base + rng.uniform(-0.5, 0.5, base.shape)) / 32768.0).astype(np.float32)
    for kind in ("ms_float", "projection"):
        out = []
        for L in (R, E):
            err = ci()
            if kind == "ms_float":
                L.opus_multistream_encoder_create.restype = vp
                L.opus_multistream_encoder_create.argtypes = [ci, ci, ci, ci, ctypes.c_char_p, ci, ctypes.POINTER(ci)]
                e = L.opus_multistream_encoder_create(Fs, 4, 3, 1, bytes([0, 1, 2, 3]), 2049, ctypes.byref(err))
                ctl, enc, destroy = L.opus_multistream_encoder_ctl, L.opus_multistream_encode_float, L.opus_multistream_encoder_destroy
            else:
                streams, coupled = ci(), ci()
                L.opus_projection_ambisonics_encoder_create.restype = vp
                L.opus_projection_ambisonics_encoder_create.argtypes = [ci, ci, ci, ctypes.POINTER(ci), ctypes.POINTER(ci), ci, ctypes.POINTER(ci)]
                e = L.opus_projection_ambisonics_encoder_create(Fs, 4, 3, ctypes.byref(streams), ctypes.byref(coupled), 2049, ctypes.byref(err))
                ctl, enc, destroy = L.opus_projection_encoder_ctl, L.opus_projection_encode_float, L.opus_projection_encoder_destroy
            assert e and err.value == 0, (kind, err.value)
            ctl.argtypes = [vp, ci, ci]; enc.argtypes = [vp, vp, ci, vp, ci]; destroy.argtypes = [vp]; destroy.restype = None
            assert ctl(e, 4010, 10) == 0 and ctl(e, 4002, 160000) == 0
            if L is E: assert ctl(e, FLOAT_ANALYSIS, 1) == 0
            buf = (ctypes.c_ubyte * 4000)(); seq = []
            for i in range(28):
                p = np.ascontiguousarray(fine[i * 960:(i + 1) * 960])
                n = enc(e, p.ctypes.data, 960, buf, 4000); seq.append(bytes(buf[:max(n, 0)]) if n > 0 else n)
            destroy(e); out.append(seq)
        bad = [i for i in range(28) if out[0][i] != out[1][i]]
        assert not bad, (kind, bad[:5], [type(x) for x in out[1][:3]])

def _loud_then_quiet(Fs, ch, n_loud, n_quiet, seed, quiet=0.004, gaps=()):
    x = music(Fs, ch, n_loud + n_quiet, seed).astype(np.float64)
    rng = np.random.default_rng(seed)
    x[n_loud:] = x[n_loud:] * quiet + (rng.integers(-3, 4, x[n_loud:].shape) if quiet > 0 else 0)
    for a, b in gaps: x[a:b] = 0
    return np.ascontiguousarray(x.astype(np.int16))

def test_lowdelay_dtx_is_only_taken_on_analysed_or_silent_frames():
    """the generalised DTX of a CELT-only encoder (src/opus_encoder.c:2565) is gated by SILK's DTX flag, which is on whenever the frame is neither analysed nor digitally
    silent (:1461): below complexity 10 or below 16 kHz a quiet-but-not-silent passage after a loud one must NOT turn into one-byte packets, whatever its energy against
    the tracked peak (found by the call-by-call comparison of the reference's fuzz_encoder_settings between the two libraries: tools/encode_trace_shim.c)"""
    for Fs, ch, fr, ctl in ((12000, 2, 240, dict(complexity=4, vbr=0, bitrate=-1, lsb_depth=8)), (48000, 2, 960, dict(complexity=5, lsb_depth=8)), (16000, 1, 320, dict(complexity=9)),
                            (48000, 1, 3840, dict(complexity=5, signal=3001, prediction_disabled=1))):
        n = 60 * (fr if fr < 1000 else 960)
        m = run(Fs, ch, 2051, [fr] * (n // fr), sig=_loud_then_quiet(Fs, ch, n // 3, n - n // 3, 31), dtx=1, **ctl)
        assert "d" not in m, (Fs, m)
    m = run(48000, 2, 2051, [960] * 40, sig=_loud_then_quiet(48000, 2, 960 * 10, 960 * 30, 32, quiet=0.0), complexity=5, dtx=1)       # digital silence does
    assert "d" in m
    # the same against the build without the float API (:1463: digital silence alone lets the generalised DTX run)
    for Fs, ch, fr, cx in ((48000, 2, 960, 10), (12000, 1, 240, 3)):
        a = capi.Enc("ref", Fs, ch, 2051, dtx=1, complexity=cx); b = capi.Enc(WHICH, Fs, ch, 2051, dtx=1, complexity=cx)
        b.L.opus_encoder_ctl.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]; assert b.L.opus_encoder_ctl(b.st, FLOAT_ANALYSIS, 0) == 0
        sig = _loud_then_quiet(Fs, ch, 15 * fr, 45 * fr, 35); sig[40 * fr:] = 0
        got = [(a.encode(sig[i * fr:(i + 1) * fr], fr), b.encode(sig[i * fr:(i + 1) * fr], fr)) for i in range(60)]
        assert all(x == y for x, y in got), [i for i, (x, y) in enumerate(got) if x != y][:5]
        assert all(x[1] > 2 for x, _ in got[:40]) and any(x[1] == 1 for x, _ in got[40:])

def test_lowdelay_dtx_sees_the_peak_tracked_before_it_was_switched_on():
    """the peak signal energy (:1310) follows the input whether or not DTX is on; switched on later, the loud-noise exemption of an analysed inactive frame (:1920) compares
    against that history.  Multi-frame calls: the digital-silence flag is the call's (a silent 20 ms inside a sounding 80 ms call is not 'silent')"""
    Fs, ch = 48000, 1
    sig = sig_for(Fs, ch, 960 * 150, 22).copy()
    sig[960 * 60:] = (sig[960 * 60:].astype(np.int32) // 512).astype(np.int16)                   # near-silence after a loud minute: the activity probability drops
    m = run(Fs, ch, 2051, [960] * 148, {70: dict(dtx=1)}, sig=sig, bitrate=64000)
    assert "d" in m[70:], m
    gaps = sig.copy()
    for k in range(25): gaps[960 * 50 + k * 3840:960 * 50 + k * 3840 + 1400] = 0
    run(Fs, ch, 2051, [3840] * 36, {5: dict(dtx=1)}, sig=gaps, bitrate=96000)
    run(Fs, ch, 2049, [2880] * 48, {5: dict(dtx=1)}, sig=gaps, bitrate=40000)


# ---- the caller's look-ahead: OPUS_SET_EXPERT_FRAME_DURATION shorter than the buffer handed to opus_encode (src/opus_encoder.c:1247, :2662-2690; run_analysis src/analysis.c:954) ----
@pytest.mark.parametrize("Fs,ch,app,dur,buf_ms,ctl", [
    (48000, 2, 2049, 5003, 20, dict(bitrate=96000, complexity=10)),                       # 10 ms frames out of 20 ms buffers, AUDIO (automatic mode)
    (48000, 1, 2048, 5004, 60, dict(bitrate=24000, complexity=10)),                       # 20 ms out of 60 ms, VOIP
    (48000, 2, 2051, 5002, 20, dict(bitrate=128000, complexity=10)),                      # 5 ms out of 20 ms, RESTRICTED_LOWDELAY (the CELT-only record)
    (16000, 1, 2048, 5003, 40, dict(bitrate=20000, complexity=10, force_mode=1000)),      # SILK-only 10 ms out of 40 ms
    (48000, 2, 2049, 5005, 120, dict(bitrate=64000, complexity=10))])                     # 40 ms (two coded frames) out of 120 ms: the analysis buffer's wrap-around guard (:964)
def test_expert_frame_duration_with_a_longer_buffer_analyses_the_whole_buffer(Fs, ch, app, dur, buf_ms, ctl):
    a, b = pair(Fs, ch, app)
    for k, v in dict(ctl, expert_frame_duration=dur).items(): assert a.set(k, v) == b.set(k, v) == 0, k
    fr = {5001: Fs // 400, 5002: Fs // 200, 5003: Fs // 100, 5004: Fs // 50, 5005: Fs // 25}[dur]; buf = Fs * buf_ms // 1000
    x = signals.music(60, channels=ch, seed=31)[::48000 // Fs]; x = np.ascontiguousarray(x if ch == 2 else x.reshape(-1))
    pos = 0
    for i in range(40):
        p, q = a.encode(x[pos:pos + buf], buf), b.encode(x[pos:pos + buf], buf)          # the buffer is `buf` samples long, the encoder codes its first `fr` (and analyses all of it)
        assert p == q, (i, p[1], q[1])
        pos += fr
        if i == 20: assert a.set("expert_frame_duration", 5000) == b.set("expert_frame_duration", 5000) == 0; fr = buf   # back to OPUS_FRAMESIZE_ARG: the pending analysis offset is consumed

def test_expert_frame_duration_look_ahead_through_the_multistream_and_float_entry_points():
    import ctypes
    vp, ci = ctypes.c_void_p, ctypes.c_int
    R = capi._proto(capi.load("ref_fxa")); E = capi._proto(capi.load(WHICH))
    encs = []
    for L in (R, E):
        L.opus_multistream_surround_encoder_create.restype = vp; L.opus_multistream_surround_encoder_create.argtypes = [ctypes.c_int32, ci, ci, vp, vp, vp, ci, vp]
        s, c, m, err = ci(), ci(), (ctypes.c_ubyte * 256)(), ci()
        e = L.opus_multistream_surround_encoder_create(48000, 3, 1, ctypes.byref(s), ctypes.byref(c), m, 2049, ctypes.byref(err)); assert e and err.value == 0
        L.opus_multistream_encode_float.argtypes = [vp, vp, ci, vp, ctypes.c_int32]
        L.opus_multistream_encoder_ctl.argtypes = [vp, ci, ci]
        for req, v in ((4002, 160000), (4010, 10), (4040, 5003)): assert L.opus_multistream_encoder_ctl(e, req, v) == 0
        if L is E: assert L.opus_multistream_encoder_ctl(e, FLOAT_ANALYSIS, 1) == 0
        encs.append((L, e))
    x = (signals.music(30, channels=2, seed=5).astype(np.float32) / 32768.0); x = np.ascontiguousarray(np.concatenate([x, x[:, :1] * 0.5], 1))
    out = [(ctypes.c_ubyte * 4000)(), (ctypes.c_ubyte * 4000)()]
    for i in range(30):
        seg = np.ascontiguousarray(x[i * 480:i * 480 + 960])
        n = [L.opus_multistream_encode_float(e, seg.ctypes.data, 960, o, 4000) for (L, e), o in zip(encs, out)]
        assert n[0] == n[1] > 0 and bytes(out[0][:n[0]]) == bytes(out[1][:n[1]]), (i, n)
