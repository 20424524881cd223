"""CPU: the SILK quantiser kernel bodies (opus_amd/csrc/silk_nsq*.h — the exact device source) run on the 64-fiber wave emulator
and compared word-for-word with the oracle restatement (which tests/test_oracle_silk.py pins to the compiled reference)."""
import ctypes, os, subprocess, numpy as np, pytest
from reflib import oracle, ROOT
from silk_inputs import NSQ_STATE, NSQ_FRAME, make_cfg, fresh_state, make_frame, make_input

def _build():
    so = os.path.join(ROOT, "tests/emu/libemu_silk.so")
    srcs = [os.path.join(ROOT, "tests/emu", f) for f in ("emu_silk.cpp", "wave_emu.cpp")]
    hdrs = [os.path.join(ROOT, "opus_amd/csrc", f) for f in ("silk_lpc.h", "silk_pitch.h", "silk_resampler.h", "silk_tables.h", "silk_nsq.h", "silk_nsq_dd.h", "silk_frame.h", "silk_host.h", "fx.h")] + [os.path.join(ROOT, "tests/emu/wave_emu.h")]
    hdrs = [h for h in hdrs if os.path.exists(h)]
    import fcntl
    with open(so + ".lock", "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(p) for p in srcs + hdrs):
            subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-rdynamic", "-I" + os.path.join(ROOT, "tests/emu"),
                                   "-I" + os.path.join(ROOT, "opus_amd/csrc")] + srcs + ["-o", so + ".tmp"])
            os.replace(so + ".tmp", so)
    return ctypes.CDLL(so)

pytestmark = pytest.mark.skipif(oracle() is None, reason="oracle lib not built")
def P(a): return a.ctypes.data_as(ctypes.c_void_p)
PERSISTENT = ["xq", "sLTP_shp_Q14", "sLPC_Q14", "sAR2_Q14", "sLF_AR_shp_Q14", "sDiff_shp_Q14", "lagPrev", "prev_gain_Q16"]

def compare_states(a, b, cfg, with_seed):
    mem = 20 * int(cfg[0])
    for name in PERSISTENT + (["rand_seed"] if with_seed else []):
        x, y = a[name], b[name]
        if name == "sLPC_Q14": x, y = x[:, :16], y[:, :16]
        if name in ("xq", "sLTP_shp_Q14"): x, y = x[:, :mem], y[:, :mem]
        assert np.array_equal(x, y), name

def drive(E, cfg, dd, n, frames, seed, T, **fkw):
    """n streams x `frames` frames through the emulator and the oracle; states imported from the oracle's layout at start and
    exported after every frame."""
    O = oracle(); rng = np.random.default_rng(seed)
    L = int(cfg[1]) * 5 * int(cfg[0])
    tw = E.emu_nsq_tile_words(T); ntiles = (n + T - 1) // T
    tiles = np.zeros(ntiles * tw, np.int32)
    st_or = fresh_state(n)
    # start from a non-trivial state: one oracle frame first
    for warm in range(1):
        fr = np.array([make_frame(rng, cfg, **fkw) for _ in range(n)], dtype=NSQ_FRAME)
        x = np.stack([make_input(rng, cfg, fr[s]["Gains_Q16"]) for s in range(n)])
        for s in range(n):
            p = np.zeros(L, np.int8)
            (O.oc_silk_nsq_del_dec if dd else O.oc_silk_nsq)(P(cfg), P(st_or[s:s + 1]), P(fr[s:s + 1]), P(x[s]), P(p))
    for s in range(n):
        E.emu_nsq_import(P(tiles[(s // T) * tw:]), T, s % T, P(st_or[s:s + 1]), P(cfg))
    for f in range(frames):
        fr = np.array([make_frame(rng, cfg, **fkw) for _ in range(n)], dtype=NSQ_FRAME)
        x = np.stack([make_input(rng, cfg, fr[s]["Gains_Q16"]) for s in range(n)])
        p_or = np.zeros((n, L), np.int8); fr_or = fr.copy()
        for s in range(n):
            (O.oc_silk_nsq_del_dec if dd else O.oc_silk_nsq)(P(cfg), P(st_or[s:s + 1]), P(fr_or[s:s + 1]), P(x[s]), P(p_or[s]))
        p_em = np.full((n, L), 99, np.int8); seeds = np.full(n, 99, np.int8)
        if dd: E.emu_silk_nsq_dd(P(cfg), P(tiles), P(fr), P(x), P(p_em), P(seeds), n)
        else: E.emu_silk_nsq(P(cfg), P(tiles), P(fr), P(x), P(p_em), n)
        bad = np.nonzero((p_em != p_or).any(axis=1))[0]
        assert len(bad) == 0, (f, bad[:8], [np.nonzero(p_em[b] != p_or[b])[0][:4] for b in bad[:4]])
        if dd: assert np.array_equal(seeds, fr_or["Seed"])
        st_em = np.zeros(n, dtype=NSQ_STATE)
        for s in range(n):
            E.emu_nsq_export(P(tiles[(s // T) * tw:]), T, s % T, P(st_em[s:s + 1]), P(cfg))
        compare_states(st_em, st_or, cfg, with_seed=not dd)

@pytest.mark.parametrize("fs,nb,shaping", [(16, 4, 24), (16, 4, 16), (8, 4, 12), (12, 2, 14), (16, 2, 20)])
def test_emu_nsq(fs, nb, shaping):
    E = _build()
    cfg = make_cfg(fs, nb, shaping, 1, False)
    drive(E, cfg, False, n=70, frames=3, seed=fs + nb + shaping, T=64)

@pytest.mark.parametrize("fs,nb,shaping,states,warp", [(16, 4, 24, 4, True), (16, 4, 24, 2, True), (16, 4, 16, 3, True), (16, 2, 24, 4, True),
                                                        (8, 4, 12, 4, True), (12, 4, 14, 2, False), (16, 4, 20, 1, True)])
def test_emu_nsq_del_dec(fs, nb, shaping, states, warp):
    E = _build()
    cfg = make_cfg(fs, nb, shaping, states, warp)
    drive(E, cfg, True, n=21, frames=3, seed=fs + nb + shaping + states, T=16)

def test_emu_generic_order_instantiation():
    """the runtime-shaping-order instantiation (orders outside the complexity table) on orders that normally take a specialised one"""
    E = _build()
    E.emu_nsq_force_generic(1)
    try:
        drive(E, make_cfg(16, 4, 24, 4, True), True, n=17, frames=2, seed=5, T=16)
        drive(E, make_cfg(16, 4, 16, 1, False), False, n=65, frames=2, seed=6, T=64)
        drive(E, make_cfg(16, 4, 18, 3, True), True, n=17, frames=2, seed=7, T=16)
    finally:
        E.emu_nsq_force_generic(0)

def test_emu_short_lags_forwarding():
    """pitch lags of 2..2.6 ms: decisionDelay == lag-3, the case where the LTP tap of sample i+1 is the entry committed at sample i
    (register forwarding in the software-pipelined loop)"""
    E = _build()
    drive(E, make_cfg(16, 4, 24, 4, True), True, n=18, frames=3, seed=11, T=16, voiced=True, max_lag_ms=2.6)
    drive(E, make_cfg(8, 4, 12, 3, True), True, n=18, frames=3, seed=12, T=16, voiced=True, max_lag_ms=2.6)
    drive(E, make_cfg(16, 4, 16, 1, False), False, n=66, frames=2, seed=13, T=64, voiced=True, max_lag_ms=2.6)

@pytest.mark.parametrize("d,length", [(16, 672), (10, 336), (16, 16), (6, 7), (12, 1024), (8, 333)])
def test_emu_lpc_analysis_filter(d, length):
    E = _build(); O = oracle(); rng = np.random.default_rng(d * 1000 + length); n = 5
    x = rng.integers(-32768, 32768, (n, length)).astype(np.int16)
    B = rng.integers(-4096, 4096, (n, d)).astype(np.int16); B[0] = rng.integers(-32768, 32768, d)          # signal 0 wraps and saturates
    got = np.full((n, length), 77, np.int16); want = got.copy()
    E.emu_silk_lpc_analysis_filter(n, P(got), P(x), P(B), length, d)
    for s in range(n): O.oc_silk_lpc_analysis_filter(P(want[s]), P(x[s]), P(B[s]), length, d)
    assert np.array_equal(got, want)

from test_oracle_silk import RS_STATE, RS_PAIRS
RS_CFG = ["resampler_function", "batchSize", "invRatio_Q16", "FIR_Order", "FIR_Fracs", "Fs_in_kHz", "Fs_out_kHz", "inputDelay", "coefs_id"]

@pytest.mark.parametrize("fs_in,fs_out,for_enc", RS_PAIRS)
def test_emu_resampler(fs_in, fs_out, for_enc):
    """the lane-per-channel streaming resampler against the oracle over consecutive calls, incl. the carried filter state"""
    E = _build(); O = oracle(); rng = np.random.default_rng(fs_in // 1000 * 100 + fs_out // 1000 + for_enc); n = 67
    st = np.zeros(n, dtype=RS_STATE)
    for s in range(n): assert O.oc_silk_resampler_init(P(st[s:s + 1]), fs_in, fs_out, for_enc) == 0
    cfg = np.array([st[0][k] for k in RS_CFG], np.int32)
    rows = np.zeros((90, n), np.int32)
    ki, ko = fs_in // 1000, fs_out // 1000
    for call in range(5):
        ms = int(rng.choice([1, 2, 10, 20])) if call else 20
        x = np.clip(np.round(rng.standard_normal((n, ki * ms)) * (30000 if call == 3 else 8000)), -32768, 32767).astype(np.int16)
        want = np.zeros((n, ko * ms), np.int16); got = np.full((n, ko * ms), 77, np.int16)
        for s in range(n): O.oc_silk_resampler(P(st[s:s + 1]), P(want[s]), P(x[s]), ki * ms)
        E.emu_silk_resampler(P(cfg), P(rows), n, P(x), ki * ms, P(got), ko * ms)
        assert np.array_equal(got, want), call
        assert np.array_equal(rows[0:6].T, st["sIIR"])
        if cfg[0] == 3: assert np.array_equal(rows[6:6 + cfg[3]].T, st["sFIR"][:, :cfg[3]])
        if cfg[0] == 2: assert np.array_equal(rows[6:14].T.astype(np.int16), st["sFIR"].view(np.int16).reshape(n, 72)[:, :8])
        assert np.array_equal(rows[42:42 + cfg[7]].T.astype(np.int16), st["delayBuf"][:, :cfg[7]])

from silk_inputs import make_pitch_frame
PE_IN = np.dtype([("prevLag", "<i4"), ("LTPCorr_Q15", "<i4"), ("search_thres1_Q16", "<i4"), ("search_thres2_Q13", "<i4")])
PE_OUT = np.dtype([("pitch", "<i4", 4), ("LTPCorr_Q15", "<i4"), ("lagIndex", "<i2"), ("contourIndex", "i1"), ("unvoiced", "i1")], align=True)

def pitch_case(rng, fs, nb, n):
    flen = (20 + 5 * nb) * fs
    x = np.zeros((n, flen), np.int16); pin = np.zeros(n, PE_IN)
    for s in range(n):
        x[s], _ = make_pitch_frame(rng, fs, nb)
        pin[s] = (int(rng.choice([0, 0, rng.integers(2 * fs, 18 * fs)])), int(rng.integers(0, 30000)), int(rng.uniform(0.6, 0.85) * 65536), int(rng.uniform(0.1, 0.6) * 8192))
    return x, pin

def pitch_oracle(x, pin, fs, cx, nb):
    O = oracle(); n = len(x); out = np.zeros(n, PE_OUT)
    O.oc_silk_pitch_analysis_core.restype = ctypes.c_int
    for s in range(n):
        pitch = np.zeros(4, np.int32); li = np.zeros(1, np.int16); ci = np.zeros(1, np.int8); lc = np.array([pin[s]["LTPCorr_Q15"]], np.int32)
        v = O.oc_silk_pitch_analysis_core(P(x[s]), P(pitch), P(li), P(ci), P(lc), int(pin[s]["prevLag"]), int(pin[s]["search_thres1_Q16"]), int(pin[s]["search_thres2_Q13"]), fs, cx, nb)
        out[s] = (pitch, lc[0], li[0], ci[0], v)
    return out

@pytest.mark.parametrize("fs,nb,cx", [(16, 4, 2), (16, 4, 0), (12, 4, 1), (8, 4, 2), (8, 4, 0), (16, 2, 2), (12, 2, 0), (8, 2, 1)])
def test_emu_pitch_analysis(fs, nb, cx):
    E = _build(); rng = np.random.default_rng(fs * 100 + nb * 10 + cx); n = 60
    x, pin = pitch_case(rng, fs, nb, n)
    want = pitch_oracle(x, pin, fs, cx, nb)
    got = np.zeros(n, PE_OUT)
    E.emu_silk_pitch(n, P(x), P(pin), P(got), fs, cx, nb)
    bad = [s for s in range(n) if got[s].tobytes() != want[s].tobytes()]
    assert not bad, (bad[:5], got[bad[0]], want[bad[0]])
    assert (want["unvoiced"] == 0).sum() > 8
