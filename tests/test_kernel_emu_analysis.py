"""CPU: the device source of the tonality / music / bandwidth analysis (opus_amd/csrc/opus_analysis.h) on the wave emulator against the compiled reference's
run_analysis (oracle/_ref/libopus_ref_fxa.so through oracle/ref_expose_fxa/x_analysis.c: the fixed-point build with the float API), call by call and field
by field: every float of the AnalysisInfo bit for bit, the detected bandwidth, the 19 leak boosts.  Rates 48 / 24 / 16 kHz, mono and stereo, frame sizes from
2.5 to 120 ms (calls that carry no, one or several analysis frames), music, speech, noise bursts, digital silence, lsb depths."""
import ctypes, os, subprocess, fcntl, numpy as np, pytest
import signals
from reflib import ref_fx, ROOT
from test_kernel_emu_silkdec import speechy
pytestmark = pytest.mark.skipif(ref_fx() is None or not os.path.exists(os.path.join(ROOT, "oracle/_ref/libref_expose_fxa.so")), reason="oracle/_ref not built")

def build_emu():
    so = os.path.join(ROOT, "tests/emu/libemu_analysis.so")
    srcs = [os.path.join(ROOT, "tests/emu", f) for f in ("emu_analysis.cpp", "wave_emu.cpp")]
    hd = os.path.join(ROOT, "opus_amd/csrc")
    deps = srcs + [os.path.join(hd, f) for f in os.listdir(hd)] + [os.path.join(ROOT, "tests/emu/wave_emu.h")]
    with open(so + ".lock", "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(p) for p in deps):
            subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-rdynamic", "-Wno-unknown-pragmas", "-I" + os.path.join(ROOT, "tests/emu"), "-I" + hd] + srcs + ["-o", so + ".tmp"])
            os.replace(so + ".tmp", so)
    E = ctypes.CDLL(so)
    E.emu_analysis_frame.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    E.emu_analysis_frame.restype = None
    return E

def ref_lib():
    from reflib import ref_expose_fxa
    X = ref_expose_fxa()
    X.ref_analysis_state_size.restype = ctypes.c_int
    X.ref_analysis_init.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int]; X.ref_analysis_init.restype = None
    X.ref_analysis_frame.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int32, ctypes.c_int, ctypes.c_void_p]; X.ref_analysis_frame.restype = None
    return X

NAMES = ["valid", "tonality", "tonality_slope", "noisiness", "activity", "music_prob", "music_prob_min", "music_prob_max", "bandwidth", "activity_probability", "max_pitch_ratio"] + ["leak%d" % i for i in range(19)]

def compare(sig, Fs, ch, frames, lsb_depth=16):
    """frames: list of frame sizes (samples at Fs), consumed one call each"""
    E, X = build_emu(), ref_lib()
    rs = ctypes.create_string_buffer(X.ref_analysis_state_size()); X.ref_analysis_init(rs, Fs, 2049)
    es = np.zeros(E.emu_analysis_state_size() // 4, np.int32)
    pos = 0; nvalid = 0
    for k, n in enumerate(frames):
        x = np.ascontiguousarray(sig[pos:pos + n], np.int16); pos += n
        assert len(x) == n, "signal too short"
        a = np.zeros(30, np.float32); b = np.zeros(30, np.float32)
        X.ref_analysis_frame(rs, x.ctypes.data, n, ch, Fs, lsb_depth, a.ctypes.data)
        E.emu_analysis_frame(es.ctypes.data, x.ctypes.data, None, n, n, ch, Fs, lsb_depth, b.ctypes.data)
        if a[0]:
            nvalid += 1
            d = np.nonzero(a.view(np.uint32) != b.view(np.uint32))[0]
            assert len(d) == 0, (Fs, ch, "call", k, "frame", n, [(NAMES[i], float(a[i]), float(b[i])) for i in d[:6]])
        else: assert b[0] == 0, (Fs, ch, k)
    return nvalid

def _sig(kind, Fs, ch, seconds, seed):
    n48 = int(48000 * seconds)
    if kind == "music": s = signals.music(n48 // 960 + 1, seed=seed)
    elif kind == "speech": s = speechy(n48 // 960 + 1, 2, seed, 960)
    else: s = signals.noise_bursts(n48 // 960 + 1, seed=seed) if hasattr(signals, "noise_bursts") else signals.music(n48 // 960 + 1, seed=seed)
    s = np.ascontiguousarray(s[::48000 // Fs])
    return s if ch == 2 else np.ascontiguousarray(s[:, :1])

@pytest.mark.parametrize("Fs,ch,kind", [(48000, 2, "music"), (48000, 1, "speech"), (24000, 2, "speech"), (16000, 1, "speech"), (16000, 2, "music"), (24000, 1, "music")])
def test_analysis_20ms_frames(Fs, ch, kind):
    n = compare(_sig(kind, Fs, ch, 3.0, 3), Fs, ch, [Fs // 50] * 140)
    assert n > 130

@pytest.mark.parametrize("Fs", [48000, 16000])
def test_analysis_frame_sizes(Fs):
    rng = np.random.default_rng(Fs)
    sizes = [Fs // 400, Fs // 200, Fs // 100, Fs // 50, Fs // 25, 3 * Fs // 50, 4 * Fs // 50, 5 * Fs // 50, 6 * Fs // 50]
    frames = [int(rng.choice(sizes)) for _ in range(120)]
    compare(_sig("music", Fs, 2, sum(frames) / Fs + 0.2, 5), Fs, 2, frames)
    compare(_sig("speech", Fs, 1, 1.5, 6), Fs, 1, [Fs // 400] * 300)

def test_analysis_silence_and_depth():
    Fs = 48000
    s = _sig("music", Fs, 2, 3.0, 9).copy()
    s[20 * 960:40 * 960] = 0                         # digital silence: the previous analysis is repeated
    s[70 * 960:75 * 960] //= 256                     # a quiet stretch: the noise-floor side of the bandwidth detector
    compare(s, Fs, 2, [960] * 120)
    compare(_sig("speech", Fs, 1, 2.0, 10) // 64, Fs, 1, [960] * 90, lsb_depth=10)
    compare(_sig("music", Fs, 2, 2.0, 11), Fs, 2, [960] * 90, lsb_depth=24)
