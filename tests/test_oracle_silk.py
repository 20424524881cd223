"""CPU: the plain-C restatement of the SILK building blocks (oracle/oc_silk_*.c) pinned against the compiled, unmodified
reference (silk_NSQ_c, silk_NSQ_del_dec_c, silk_LPC_analysis_filter, DIV32/INVERSE32_varQ).  Bit-exact or fail."""
import ctypes, numpy as np, pytest
from reflib import ref_expose, oracle
from silk_inputs import NSQ_STATE, NSQ_FRAME, make_cfg, fresh_state, make_frame, make_input

pytestmark = pytest.mark.skipif(ref_expose() is None or oracle() is None, reason="oracle/_ref or oracle lib not built")
def P(a): return a.ctypes.data_as(ctypes.c_void_p)

def test_varq_division():
    X, O = ref_expose(), oracle(); rng = np.random.default_rng(0)
    for f in (X.ref_silk_div32_varQ, X.ref_silk_inverse32_varQ, O.oc_silk_div32_varQ, O.oc_silk_inverse32_varQ): f.restype = ctypes.c_int32
    for _ in range(20000):
        a = int(rng.integers(-2**31 + 1, 2**31 - 1)); b = int(2 ** rng.uniform(0, 31)) * int(rng.choice([-1, 1])) or 1
        q = int(rng.integers(0, 32))
        assert X.ref_silk_div32_varQ(a, b, q) == O.oc_silk_div32_varQ(a, b, q), (a, b, q)
        q = int(rng.integers(1, 48))
        assert X.ref_silk_inverse32_varQ(b, q) == O.oc_silk_inverse32_varQ(b, q), (b, q)

def test_lpc_analysis_filter():
    X, O = ref_expose(), oracle(); rng = np.random.default_rng(1)
    for it in range(300):
        d = int(rng.choice([6, 8, 10, 12, 16])); n = int(rng.integers(d, 700))
        x = rng.integers(-32768, 32768, n).astype(np.int16)
        B = rng.integers(-32768 if it % 3 == 0 else -4096, 32768 if it % 3 == 0 else 4096, d).astype(np.int16)   # 1/3 of the cases wrap and saturate
        o1 = np.full(n, 77, np.int16); o2 = o1.copy()
        X.ref_silk_lpc_analysis_filter(P(o1), P(x), P(B), n, d); O.oc_silk_lpc_analysis_filter(P(o2), P(x), P(B), n, d)
        assert np.array_equal(o1, o2)

def run_ref(cfg, dd, st, fr, x):
    X = ref_expose()
    pulses = np.zeros(len(x), np.int8)
    ind = np.array([fr["signalType"], fr["quantOffsetType"], fr["NLSFInterpCoef_Q2"], fr["Seed"]], np.int8)
    hs, ti, lf, ga, pl = (np.ascontiguousarray(fr[k]) for k in ("HarmShapeGain_Q14", "Tilt_Q14", "LF_shp_Q14", "Gains_Q16", "pitchL"))
    pc, lt, ar = (np.ascontiguousarray(fr[k]) for k in ("PredCoef_Q12", "LTPCoef_Q14", "AR_Q13"))
    X.ref_silk_nsq(P(cfg), int(dd), P(st), P(ind), P(x), P(pulses), P(pc), P(lt), P(ar), P(hs), P(ti), P(lf), P(ga), P(pl),
                   int(fr["Lambda_Q10"]), int(fr["LTP_scale_Q14"]))
    return pulses, int(ind[3])

def run_oracle(cfg, dd, st, fr, x):
    O = oracle()
    pulses = np.zeros(len(x), np.int8)
    f = np.array([fr], dtype=NSQ_FRAME)
    (O.oc_silk_nsq_del_dec if dd else O.oc_silk_nsq)(P(cfg), P(st), P(f), P(x), P(pulses))
    return pulses, int(f[0]["Seed"])

def state_equal(a, b, cfg):
    """every persistent word; sLPC_Q14[16:] is per-call scratch in the reference (only [0:16] carries over)"""
    for name in NSQ_STATE.names:
        x, y = a[name][0], b[name][0]
        if name == "sLPC_Q14": x, y = x[:16], y[:16]
        if name in ("xq", "sLTP_shp_Q14"): x, y = x[:20 * int(cfg[0])], y[:20 * int(cfg[0])]   # [ltp_mem:] is dead after the memmove
        if name == "rand_seed": continue       # silk_NSQ_del_dec_c never touches it; silk_NSQ_c leaves the running dither there, checked below
        if not np.array_equal(x, y): return name
    return None

CASES = [(16, 4, 24, 1, False, False), (16, 4, 16, 1, False, False), (8, 4, 12, 1, False, False), (12, 2, 14, 1, False, False),
         (16, 4, 24, 4, True, True), (16, 4, 24, 2, True, True), (16, 4, 16, 3, True, True), (16, 2, 24, 4, True, True),
         (8, 4, 12, 4, True, True), (12, 4, 14, 2, False, True), (16, 4, 20, 1, True, True)]

@pytest.mark.needs_ref
@pytest.mark.parametrize("fs,nb,shaping,states,warp,dd", CASES)
def test_nsq_matches_reference(fs, nb, shaping, states, warp, dd):
    cfg = make_cfg(fs, nb, shaping, states, warp)
    rng = np.random.default_rng(fs * 1000 + nb * 100 + shaping + states + dd)
    for stream in range(6):
        s_ref, s_or = fresh_state(), fresh_state()
        nz = 0
        for frame in range(8):
            fr = make_frame(rng, cfg)
            x = make_input(rng, cfg, fr["Gains_Q16"])
            p1, seed1 = run_ref(cfg, dd, s_ref, fr, x)
            p2, seed2 = run_oracle(cfg, dd, s_or, fr, x)
            assert np.array_equal(p1, p2), (stream, frame, np.nonzero(p1 != p2)[0][:8])
            assert seed1 == seed2
            bad = state_equal(s_ref, s_or, cfg)
            assert bad is None, (stream, frame, bad)
            if not dd: assert s_ref["rand_seed"][0] == s_or["rand_seed"][0]
            nz += int(np.count_nonzero(p1))
        assert nz > 100                       # the synthetic inputs do exercise the quantiser

RS_STATE = np.dtype([("sIIR", "<i4", 6), ("sFIR", "<i4", 36), ("delayBuf", "<i2", 96), ("resampler_function", "<i4"), ("batchSize", "<i4"),
                     ("invRatio_Q16", "<i4"), ("FIR_Order", "<i4"), ("FIR_Fracs", "<i4"), ("Fs_in_kHz", "<i4"), ("Fs_out_kHz", "<i4"),
                     ("inputDelay", "<i4"), ("coefs_id", "<i4")], align=True)
RATES = [8000, 12000, 16000, 24000, 48000]
RS_PAIRS = [(i, o, 1) for i in RATES for o in (8000, 12000, 16000)] + [(i, o, 0) for i in (8000, 12000, 16000) for o in RATES]
RS_FIELDS = ["resampler_function", "batchSize", "invRatio_Q16", "FIR_Order", "FIR_Fracs", "Fs_in_kHz", "Fs_out_kHz", "inputDelay"]

@pytest.mark.needs_ref
@pytest.mark.parametrize("fs_in,fs_out,for_enc", RS_PAIRS)
def test_resampler_matches_reference(fs_in, fs_out, for_enc):
    """silk_resampler_init + silk_resampler (silk/resampler.c:79,:183) over consecutive calls of mixed lengths"""
    X, O = ref_expose(), oracle(); rng = np.random.default_rng(fs_in // 1000 * 100 + fs_out // 1000 + for_enc)
    ref = np.zeros(X.ref_silk_resampler_state_size(), np.uint8); st = np.zeros(1, dtype=RS_STATE)
    assert X.ref_silk_resampler_init(P(ref), fs_in, fs_out, for_enc) == 0 and O.oc_silk_resampler_init(P(st), fs_in, fs_out, for_enc) == 0
    ki, ko = fs_in // 1000, fs_out // 1000
    for call in range(12):
        ms = int(rng.choice([1, 2, 5, 10, 20]))
        amp = 32767 if call % 4 == 3 else 8000                   # every fourth call saturates
        x = np.clip(np.round(rng.standard_normal(ki * ms) * amp), -32768, 32767).astype(np.int16)
        o1 = np.full(ko * ms, 77, np.int16); o2 = o1.copy()
        X.ref_silk_resampler(P(ref), P(o1), P(x), len(x)); O.oc_silk_resampler(P(st), P(o2), P(x), len(x))
        assert np.array_equal(o1, o2), (call, ms)
        assert bytes(ref[:360]) == st.tobytes()[:360], call      # sIIR, sFIR, delayBuf: identical layout in both structs
    for i, k in enumerate(RS_FIELDS):
        assert int(np.frombuffer(bytes(ref[360 + 4 * i:364 + 4 * i]), np.int32)[0]) == int(st[k][0]), k

def test_resampler_rejects_unsupported_rates():
    O = oracle(); st = np.zeros(1, dtype=RS_STATE)
    assert O.oc_silk_resampler_init(P(st), 44100, 16000, 1) == -1 and O.oc_silk_resampler_init(P(st), 48000, 24000, 1) == -1
    assert O.oc_silk_resampler_init(P(st), 24000, 48000, 0) == -1

from silk_inputs import make_pitch_frame
def _pitch_both(x, prev_lag, ltpcorr, t1, t2, fs, cx, nb):
    from reflib import ref_fx
    R, O = ref_fx(), oracle()
    out = []
    for fn, extra in ((R.silk_pitch_analysis_core, (0,)), (O.oc_silk_pitch_analysis_core, ())):
        pitch = np.full(4, -7, np.int32); li = np.zeros(1, np.int16); ci = np.zeros(1, np.int8); lc = np.array([ltpcorr], np.int32)
        fn.restype = ctypes.c_int
        v = fn(P(x), P(pitch), P(li), P(ci), P(lc), int(prev_lag), int(t1), int(t2), fs, cx, nb, *extra)
        out.append((v, pitch[:nb].tolist(), int(li[0]), int(ci[0]), int(lc[0])))
    return out

@pytest.mark.needs_ref
@pytest.mark.parametrize("fs,nb", [(16, 4), (12, 4), (8, 4), (16, 2), (12, 2), (8, 2)])
def test_pitch_analysis_matches_reference(fs, nb):
    """silk_pitch_analysis_core (silk/fixed/pitch_analysis_core_FIX.c:82): voicing decision, lags, lagIndex, contourIndex, LTPCorr"""
    rng = np.random.default_rng(fs * 10 + nb); voiced = 0
    for it in range(160):
        x, kind = make_pitch_frame(rng, fs, nb)
        cx = int(rng.integers(0, 3)); prev = int(rng.choice([0, 0, rng.integers(2 * fs, 18 * fs)])); ltp = int(rng.integers(0, 30000))
        t1 = int(rng.uniform(0.6, 0.85) * 65536); t2 = int(rng.uniform(0.1, 0.6) * 8192)
        a, b = _pitch_both(x, prev, ltp, t1, t2, fs, cx, nb)
        assert a == b, (it, kind, cx, prev, a, b)
        voiced += a[0] == 0
    assert voiced > 25              # the synthetic frames do reach stages 2 and 3

@pytest.mark.needs_ref
def test_pitch_helpers_match_reference():
    from reflib import ref_fx
    R, O = ref_fx(), oracle(); rng = np.random.default_rng(5)
    R.silk_lin2log.restype = ctypes.c_int32; O.oc_silk_lin2log.restype = ctypes.c_int32
    for v in list(range(1, 600)) + [int(2 ** rng.uniform(0, 31)) for _ in range(2000)]:
        assert R.silk_lin2log(v) == O.oc_silk_lin2log(v), v
    for it in range(100):
        n = int(rng.integers(2, 700)); x = (rng.standard_normal(n) * 2 ** rng.uniform(0, 15)).clip(-32768, 32767).astype(np.int16)
        e1, s1, e2, s2 = ctypes.c_int32(), ctypes.c_int(), ctypes.c_int32(), ctypes.c_int()
        R.silk_sum_sqr_shift(ctypes.byref(e1), ctypes.byref(s1), P(x), n); O.oc_silk_sum_sqr_shift(ctypes.byref(e2), ctypes.byref(s2), P(x), n)
        assert (e1.value, s1.value) == (e2.value, s2.value)
        for fn_r, fn_o, nst, olen in ((R.silk_resampler_down2, O.oc_silk_resampler_down2, 2, n // 2), (R.silk_resampler_down2_3, O.oc_silk_resampler_down2_3, 6, 2 * (n // 3))):
            m = n - n % 6
            if m < 6: continue
            sa = rng.integers(-1000, 1000, 6).astype(np.int32); sb = sa.copy()
            oa = np.zeros(m, np.int16); ob = oa.copy()
            fn_r(P(sa), P(oa), P(x), m); fn_o(P(sb), P(ob), P(x), m)
            assert np.array_equal(oa, ob) and np.array_equal(sa[:nst], sb[:nst])
