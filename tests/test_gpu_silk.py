"""GPU: the batched SILK noise-shaping quantisers through the C ABI (opusgpu_nsq_*) against the oracle restatement (itself pinned to the
compiled reference's silk_NSQ_c / silk_NSQ_del_dec_c by tests/test_oracle_silk.py) and, where oracle/_ref is present, against the
compiled reference directly.  Bit-exact pulses, Seed and every persistent state word."""
import ctypes, numpy as np, pytest
from reflib import oracle, ref_expose
from silk_inputs import NSQ_STATE, NSQ_FRAME, make_cfg, fresh_state, make_frame, make_input

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(oracle() is None, reason="oracle lib not built")]
def P(a): return a.ctypes.data_as(ctypes.c_void_p)
PERSISTENT = ["xq", "sLTP_shp_Q14", "sLPC_Q14", "sAR2_Q14", "sLF_AR_shp_Q14", "sDiff_shp_Q14", "lagPrev", "prev_gain_Q16"]

def oracle_frame(cfg, dd, st, fr, x):
    O = oracle(); n, L = x.shape
    p = np.zeros((n, L), np.int8); fr = fr.copy()
    for s in range(n):
        (O.oc_silk_nsq_del_dec if dd else O.oc_silk_nsq)(P(cfg), P(st[s:s + 1]), P(fr[s:s + 1]), P(x[s]), P(p[s]))
    return p, fr["Seed"].copy()

def states_of(b, n):
    st = np.zeros(n, dtype=NSQ_STATE)
    for s in range(n): st[s:s + 1] = np.frombuffer(b.export_state(s), dtype=NSQ_STATE)
    return st

def check_states(a, b, cfg, with_seed):
    mem = 20 * int(cfg[0])
    for name in PERSISTENT + (["rand_seed"] if with_seed else []):
        x, y = a[name], b[name]
        if name == "sLPC_Q14": x, y = x[:, :16], y[:, :16]
        if name in ("xq", "sLTP_shp_Q14"): x, y = x[:, :mem], y[:, :mem]
        assert np.array_equal(x, y), name

CASES = [(16, 4, 24, 1, False), (16, 4, 16, 1, False), (8, 4, 12, 1, False), (12, 2, 14, 1, False),
         (16, 4, 24, 4, True), (16, 4, 24, 2, True), (16, 4, 16, 3, True), (16, 2, 24, 4, True), (8, 4, 12, 4, True), (12, 4, 14, 2, False), (16, 4, 20, 1, True)]

@pytest.mark.parametrize("fs,nb,shaping,states,warp", CASES)
def test_gpu_nsq_matches_oracle(fs, nb, shaping, states, warp):
    import opus_amd
    cfg = make_cfg(fs, nb, shaping, states, warp); dd = states > 1 or warp
    n = 150 if not dd else 70                        # > 2 tiles incl. a ragged tail
    rng = np.random.default_rng(fs * 7 + nb + shaping + states)
    b = opus_amd.NsqBatch(n, cfg)
    st = fresh_state(n)
    for f in range(4):
        fr = np.array([make_frame(rng, cfg) for _ in range(n)], dtype=NSQ_FRAME)
        x = np.stack([make_input(rng, cfg, fr[s]["Gains_Q16"]) for s in range(n)])
        p_or, seed_or = oracle_frame(cfg, dd, st, fr, x)
        p_gpu, seed_gpu = b.run(fr, x)
        bad = np.nonzero((p_gpu != p_or).any(axis=1))[0]
        assert len(bad) == 0, (f, bad[:8])
        assert np.array_equal(seed_gpu, seed_or)
        if f in (0, 3): check_states(states_of(b, n), st, cfg, with_seed=not dd)
    b.close()

def test_gpu_nsq_import_export_and_reference():
    """state migration: run 2 frames on the compiled reference, import its silk_nsq_state into the batch, continue on the GPU and on the
    reference side by side (the memcpy contract of the boundary)."""
    X = ref_expose()
    if X is None: pytest.skip("oracle/_ref not built")
    import opus_amd
    from test_oracle_silk import run_ref
    cfg = make_cfg(16, 4, 24, 4, True); n = 20; rng = np.random.default_rng(99)
    st = fresh_state(n)
    b = opus_amd.NsqBatch(n, cfg)
    for f in range(4):
        fr = np.array([make_frame(rng, cfg) for _ in range(n)], dtype=NSQ_FRAME)
        x = np.stack([make_input(rng, cfg, fr[s]["Gains_Q16"]) for s in range(n)])
        if f == 2:
            for s in range(n): b.import_state(s, st[s:s + 1].tobytes())
        p_ref = np.zeros_like(x, dtype=np.int8); seeds = np.zeros(n, np.int8)
        for s in range(n):
            p_ref[s], seeds[s] = run_ref(cfg, True, st[s:s + 1], fr[s], x[s])
        if f >= 2:
            p_gpu, seed_gpu = b.run(fr, x)
            assert np.array_equal(p_gpu, p_ref) and np.array_equal(seed_gpu, seeds)
    check_states(states_of(b, n), st, cfg, with_seed=False)
    b.close()

def test_gpu_nsq_bad_args():
    import opus_amd
    with pytest.raises(opus_amd.OpusError): opus_amd.NsqBatch(4, make_cfg(24, 4, 24, 4, True))      # fs_kHz out of range
    with pytest.raises(opus_amd.OpusError): opus_amd.NsqBatch(0, make_cfg())
    b = opus_amd.NsqBatch(3, make_cfg())
    with pytest.raises(opus_amd.OpusError): b.export_state(3)
    b.close()

@pytest.mark.parametrize("d,length", [(16, 672), (10, 336), (16, 16), (6, 7), (12, 1024), (8, 333)])
def test_gpu_lpc_analysis_filter(d, length):
    import opus_amd
    O = oracle(); rng = np.random.default_rng(d * 1000 + length); n = 300
    x = rng.integers(-32768, 32768, (n, length)).astype(np.int16)
    B = rng.integers(-4096, 4096, (n, d)).astype(np.int16); B[::7] = rng.integers(-32768, 32768, (len(B[::7]), d))     # every 7th wraps and saturates
    got = opus_amd.silk_lpc_analysis_filter(x, B)
    want = np.zeros_like(x)
    for s in range(n): O.oc_silk_lpc_analysis_filter(P(want[s]), P(x[s]), P(B[s]), length, d)
    assert np.array_equal(got, want)
    X = ref_expose()
    if X is not None:
        ref = np.zeros_like(x[:20])
        for s in range(20): X.ref_silk_lpc_analysis_filter(P(ref[s]), P(x[s]), P(B[s]), length, d)
        assert np.array_equal(got[:20], ref)

def test_gpu_lpc_analysis_filter_bad_args():
    import opus_amd
    x = np.zeros((2, 100), np.int16)
    for d in (4, 7, 18):
        with pytest.raises(opus_amd.OpusError): opus_amd.silk_lpc_analysis_filter(x, np.zeros((2, d), np.int16))
    with pytest.raises(opus_amd.OpusError): opus_amd.silk_lpc_analysis_filter(np.zeros((2, 2000), np.int16), np.zeros((2, 16), np.int16))

from test_oracle_silk import RS_STATE, RS_PAIRS

@pytest.mark.parametrize("fs_in,fs_out,for_enc", RS_PAIRS)
def test_gpu_resampler(fs_in, fs_out, for_enc):
    """opusgpu_resampler_* against the oracle (pinned to silk_resampler by tests/test_oracle_silk.py) over consecutive calls, then the state
    blob against the oracle's (same field order)"""
    import opus_amd
    O = oracle(); rng = np.random.default_rng(fs_in // 1000 * 100 + fs_out // 1000 + for_enc); n = 200
    st = np.zeros(n, dtype=RS_STATE)
    for s in range(n): assert O.oc_silk_resampler_init(P(st[s:s + 1]), fs_in, fs_out, for_enc) == 0
    b = opus_amd.ResamplerBatch(n, fs_in, fs_out, for_enc)
    ki, ko = fs_in // 1000, fs_out // 1000
    for call in range(5):
        ms = int(rng.choice([1, 2, 10, 20])) if call else 20
        x = np.clip(np.round(rng.standard_normal((n, ki * ms)) * (30000 if call == 3 else 8000)), -32768, 32767).astype(np.int16)
        want = np.zeros((n, ko * ms), np.int16)
        for s in range(n): O.oc_silk_resampler(P(st[s:s + 1]), P(want[s]), P(x[s]), ki * ms)
        assert np.array_equal(b.run(x), want), call
    for s in (0, 63, 64, n - 1):
        blob = np.frombuffer(b.export_state(s), dtype=RS_STATE)[0]
        nfir = st[s]["FIR_Order"] if st[s]["resampler_function"] == 3 else 2 if st[s]["resampler_function"] == 2 else 0      # words of FIR tail in use (8 int16 = 4 words... compared as int16 below)
        assert np.array_equal(blob["sIIR"], st[s]["sIIR"])
        if st[s]["resampler_function"] == 3: assert np.array_equal(blob["sFIR"][:nfir], st[s]["sFIR"][:nfir])
        if st[s]["resampler_function"] == 2: assert np.array_equal(blob["sFIR"].view(np.int16)[:8], st[s]["sFIR"].view(np.int16)[:8])
        d = int(st[s]["inputDelay"]); assert np.array_equal(blob["delayBuf"][:d], st[s]["delayBuf"][:d])
    # migrate channel 0's state into channel 1 and check both then produce the same output for the same input
    b.import_state(1, b.export_state(0))
    x = np.clip(np.round(rng.standard_normal((n, ki * 10)) * 8000), -32768, 32767).astype(np.int16); x[1] = x[0]
    y = b.run(x); assert np.array_equal(y[0], y[1])
    b.close()

def test_gpu_resampler_bad_args():
    import opus_amd
    with pytest.raises(opus_amd.OpusError): opus_amd.ResamplerBatch(4, 44100, 16000, 1)
    with pytest.raises(opus_amd.OpusError): opus_amd.ResamplerBatch(4, 48000, 24000, 1)
    b = opus_amd.ResamplerBatch(4, 48000, 16000, 1)
    with pytest.raises(opus_amd.OpusError): b.run(np.zeros((4, 50), np.int16))          # not a whole number of milliseconds
    with pytest.raises(opus_amd.OpusError): b.run(np.zeros((4, 24), np.int16))          # < 1 ms
    b.close()

from test_kernel_emu_silk import pitch_case, pitch_oracle

@pytest.mark.parametrize("fs,nb,cx", [(16, 4, 2), (16, 4, 0), (12, 4, 1), (8, 4, 2), (8, 4, 0), (16, 2, 2), (12, 2, 0), (8, 2, 1)])
def test_gpu_pitch_analysis(fs, nb, cx):
    """opusgpu_silk_pitch_analysis_batch against the oracle (pinned to silk_pitch_analysis_core by tests/test_oracle_silk.py): voicing, the four
    lags, lagIndex, contourIndex, LTPCorr — byte-identical output records"""
    import opus_amd
    rng = np.random.default_rng(fs * 100 + nb * 10 + cx + 1); n = 400
    x, pin = pitch_case(rng, fs, nb, n)
    want = pitch_oracle(x, pin, fs, cx, nb)
    got = opus_amd.silk_pitch_analysis(x, pin, fs, cx, nb)
    bad = [s for s in range(n) if got[s].tobytes() != want[s].tobytes()]
    assert not bad, (bad[:5], got[bad[0]], want[bad[0]])
    assert (want["unvoiced"] == 0).sum() > 60

def test_gpu_pitch_analysis_bad_args():
    import opus_amd
    pin = np.zeros(2, np.dtype([("a", "<i4", 4)]))
    with pytest.raises(opus_amd.OpusError): opus_amd.lib() and opus_amd.silk_pitch_analysis(np.zeros((2, 40 * 16), np.int16), pin, 16, 3, 4)
