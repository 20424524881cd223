"""Where the reference built with DISABLE_FLOAT_API (oracle/_ref/libopus_ref_fx.so, what the plain-C restatement follows) and the fixed-point build users get by
default (float API on: src/analysis.c + mlp.c steer the encoder, libopus_ref_fxa.so) agree and where they do not -- both sides are the compiled reference,
tools/analysis_gap.py.  The forced SILK-only (BASELINE config 3) and forced hybrid (config 4) encoders and an unforced VOIP encoder produce IDENTICAL packets with and
without the analysis; CELT-coded frames at complexity 10 (the FIXED_POINT build runs the analysis at complexity 10 only, src/opus_encoder.c:1249; config 2, unforced
AUDIO) differ.  The product runs the analysis on the device since round 3 (opus_amd/csrc/opus_analysis.h, tests/test_hostemu_analysis.py, tests/test_gpu_analysis.py);
this file keeps the map of which configurations it matters for, and pins the frame-by-frame analysis oracle (oracle/ref_expose_fxa) those suites lean on."""
import os, sys, pytest
from reflib import ref_fx, ROOT
sys.path.insert(0, os.path.join(ROOT, "tools"))
pytestmark = pytest.mark.skipif(ref_fx() is None or not os.path.exists(os.path.join(ROOT, "oracle/_ref/libopus_ref_fxa.so")), reason="oracle/_ref not built")

def test_which_configurations_the_analysis_changes():
    import analysis_gap as G
    r = {c[0].split(":")[0].split(",")[0]: G.run(*c, frames=100) for c in G.CASES}
    for k in ("config 3", "config 4", "VOIP 16 kHz mono 20 kb/s"): assert r[k]["identical"] == r[k]["frames"], r[k]
    assert r["config 2"]["identical"] < r["config 2"]["frames"] and r["config 2"]["toc_differs"] == 0, r["config 2"]      # same decisions, different allocation tuning


FIELDS = ["valid", "tonality", "tonality_slope", "noisiness", "activity", "music_prob", "music_prob_min", "music_prob_max", "bandwidth", "activity_probability", "max_pitch_ratio"]

def _analyse(sig, Fs, ch, frames):
    import ctypes, numpy as np
    from reflib import ref_expose_fxa
    X = ref_expose_fxa()
    X.ref_analysis_state_size.restype = ctypes.c_int
    X.ref_analysis_init.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int]; X.ref_analysis_init.restype = None
    X.ref_analysis_frame.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int32, ctypes.c_int, ctypes.c_void_p]; X.ref_analysis_frame.restype = None
    st = ctypes.create_string_buffer(X.ref_analysis_state_size())
    X.ref_analysis_init(st, Fs, 2049)
    n = Fs // 50; out = np.zeros((frames, 30), np.float32)
    for f in range(frames):
        x = np.ascontiguousarray(sig[f * n:(f + 1) * n], np.int16)
        X.ref_analysis_frame(st, x.ctypes.data, n, ch, Fs, 16, out[f].ctypes.data)
    return out

def test_analysis_oracle_runs_frame_by_frame():
    """the oracle the device analysis is checked against field by field (tests/test_kernel_emu_analysis.py): the compiled reference's run_analysis on the encoder's
    int16 input, one AnalysisInfo per 20 ms frame.  Pinned here on its own: deterministic, valid after the first frames, and telling this repo's music corpus from
    its speech corpus the way the encoder's mode decision needs it to."""
    import numpy as np, signals
    from test_kernel_emu_silkdec import speechy
    m = _analyse(signals.music(60, seed=3), 48000, 2, 60); m2 = _analyse(signals.music(60, seed=3), 48000, 2, 60)
    s = _analyse(speechy(60, 1, 5, 960)[::3], 16000, 1, 60)
    assert np.array_equal(m, m2)
    assert m[10:, 0].all() and s[10:, 0].all()                      # valid
    assert 12 <= m[-1, 8] <= 20 and 12 <= s[-1, 8] <= 20             # detected bandwidth, in bands
    assert np.isfinite(m).all() and np.isfinite(s).all()
    assert m[20:, 5].mean() > s[20:, 5].mean()                      # music_prob: music above speech
