"""MI355X: the CELT-only kernel pipeline (oa_encode_kernel cut before the PVQ -> oa_celt_pvq_kernel, four streams per wave -> oa_celt_back_kernel) against the compiled
reference: tests/celt_pipe_check.py's matrix with the pipeline forced for its narrow batches (a wide launch, >= 64 streams, takes it by itself: the bench's parity samples and
the full-size tests cover that), the settings fuzzers with the pipeline forced, and byte identity between the two launch forms on a 4,096-stream batch."""
import os, subprocess, sys, ctypes, numpy as np, pytest
import test_hostemu_fuzz as Z
pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))

def test_gpu_celt_pipeline_matches_the_reference():
    r = subprocess.run([sys.executable, os.path.join(HERE, "celt_pipe_check.py"), "gpu"], env=dict(os.environ, OPUS_AMD_FLOAT_ANALYSIS="0"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "16 cases" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]

def _celt_seeds(lo, n):
    """seeds of the encoder fuzzers whose encoder is a CELT-only application (OPUS_APPLICATION_RESTRICTED_LOWDELAY)"""
    out = []
    s = lo
    while len(out) < n:
        rng = np.random.default_rng(1000 + s); rng.choice([8000, 12000, 16000, 24000, 48000]); rng.choice([1, 2])
        if int(rng.choice([2048, 2049, 2051, 2051])) == 2051: out.append(s)
        s += 1
    return out

def test_settings_fuzz_celt_only_through_the_pipeline(monkeypatch):
    monkeypatch.setattr(Z, "WHICH", "gpu"); monkeypatch.setattr(Z, "PIPELINE", 1); monkeypatch.setattr(Z, "TRPRE", 1)      # ... and the transient pre-pass for these narrow launches too
    for seed in _celt_seeds(3000, 24): Z.fuzz(seed)

def test_batch_abi_fuzz_through_the_celt_pipeline(monkeypatch):
    monkeypatch.setattr(Z, "WHICH", "gpu"); monkeypatch.setattr(Z, "PIPELINE", 1); monkeypatch.setattr(Z, "TRPRE", 1)
    for seed in range(3100, 3124): Z.fuzz_batch(seed)

def test_pipeline_and_one_kernel_agree_on_a_wide_batch():
    """4,096 streams x 12 frames of the bench corpus: the pipeline (the default of a wide launch) and the one-kernel path give identical packets, final ranges and stream records"""
    import opus_amd as oa, signals
    S, T = 4096, 12
    sig = [signals.music(T + 1, seed=s) if s % 5 else signals.noise_bursts(T + 1, seed=s) for s in range(64)]
    res = []
    for mode in (0, 1):
        b = oa.EncoderBatch(S, channels=2); b.ctl(oa.OPUS_SET_BITRATE_REQUEST, 128000); b.ctl(oa.OPUS_SET_COMPLEXITY_REQUEST, 10); b.ctl(11902, mode)
        for s in range(0, S, 7): b.ctl(oa.OPUS_SET_BITRATE_REQUEST, 24000 + 4000 * (s % 31), stream=s)
        out = []
        for i in range(T):
            pcm = np.stack([sig[s % 64][i * 960:(i + 1) * 960].reshape(-1) for s in range(S)])
            pk, lens, rng = b.encode(pcm, 960); out.append((pk, lens.copy(), rng.copy()))
        st = [b.export_state(s) for s in range(0, S, 64)]
        res.append((out, st)); b.close()
    for (p0, l0, r0), (p1, l1, r1) in zip(res[0][0], res[1][0]):
        assert np.array_equal(l0, l1) and np.array_equal(r0, r1) and p0 == p1
    assert res[0][1] == res[1][1]
