"""CPU: the *kernel body source* (opus_amd/csrc/celt_enc_*.h) executed lane-by-lane on the CPU wave emulator
(tests/emu) must produce the oracle's packets, final range and carried state, frame after frame.  This is how the
GPU code is debugged in a container without a GPU; the same checks run on the real MI355X in test_gpu_parity.py."""
import ctypes, os, subprocess, numpy as np, pytest
from reflib import oracle, ROOT
import signals
from test_oracle_encoder import OracleEnc
import emu_harness as EH

def _build():
    so = os.path.join(ROOT, "tests/emu/libemu_encoder.so")
    srcs = [os.path.join(ROOT, "tests/emu", f) for f in ("emu_encoder.cpp", "wave_emu.cpp")]
    hdrs = [os.path.join(ROOT, "opus_amd/csrc", f) for f in os.listdir(os.path.join(ROOT, "opus_amd/csrc")) if f.endswith(".h")] + [os.path.join(ROOT, "tests/emu/wave_emu.h")]
    import fcntl
    with open(so + ".lock", "w") as lk:                     # pytest-xdist workers must not race the rebuild
        fcntl.flock(lk, fcntl.LOCK_EX)
        if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(p) for p in srcs + hdrs):
            subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-rdynamic", "-I" + os.path.join(ROOT, "tests/emu"),
                                   "-I" + os.path.join(ROOT, "opus_amd/csrc")] + srcs + ["-o", so + ".tmp"])
            os.replace(so + ".tmp", so)
    return ctypes.CDLL(so)

pytestmark = pytest.mark.skipif(oracle() is None, reason="oracle lib not built")

def _run(channels, sig, frame, nframes, **kw):
    E = _build()
    st = EH.new_stream(E, channels, **kw)
    oe = OracleEnc(channels, **kw)
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    for i in range(nframes):
        pcm = np.ascontiguousarray(sig[i * frame:(i + 1) * frame])
        a = oe.encode(pcm, frame)
        out = np.zeros(1500, np.uint8); ln = np.zeros(1, np.int32); rg = np.zeros(1, np.uint32)
        E.emu_encode_batch(P(st), P(pcm), 1, frame, 1276, P(out), 1500, P(ln), P(rg))
        b = (bytes(out[:max(int(ln[0]), 0)]), int(ln[0]), int(rg[0]))
        assert a == b, (i, a[1], b[1], hex(a[2]), hex(b[2]))

def test_emu_config2_stereo_128k_c10():
    _run(2, signals.music(40, seed=0), 960, 40, bitrate=128000, complexity=10)

@pytest.mark.parametrize("channels,bitrate,complexity,frame", [
    (2, 64000, 10, 960), (2, 24000, 10, 960), (2, 510000, 10, 960), (1, 64000, 10, 960), (1, 12000, 5, 960),
    (2, 96000, 5, 960), (2, 96000, 0, 960), (2, 48000, 3, 960), (2, 128000, 10, 480), (2, 128000, 10, 240),
    (2, 128000, 10, 120), (1, 48000, 10, 480), (2, 16000, 10, 960), (2, 8000, 10, 960)])
def test_emu_rates_sizes(channels, bitrate, complexity, frame):
    n = 24 * 960 // frame
    _run(channels, signals.music(24, channels=channels, seed=7), frame, min(n, 60), bitrate=bitrate, complexity=complexity)

@pytest.mark.parametrize("kind", ["bursts", "tone", "silence", "loud"])
def test_emu_signal_kinds(kind):
    sig = dict(bursts=signals.noise_bursts(40), tone=signals.tone(40, freq=997.0), silence=signals.silence_then_music(40),
               loud=signals.music(40, amp=60000.0))[kind]
    _run(2, sig, 960, 40, bitrate=128000, complexity=10)

def test_emu_ctls():
    _run(2, signals.music(24, seed=3), 960, 24, bitrate=96000, complexity=10, vbr_constraint=0)
    _run(2, signals.music(24, seed=4), 960, 24, bitrate=96000, complexity=10, force_channels=1)
    _run(2, signals.music(24, seed=5), 960, 24, bitrate=64000, complexity=10, user_bandwidth=1103)
    _run(2, signals.music(24, seed=6), 960, 24, bitrate=64000, complexity=10, max_bandwidth=1104, disable_inv=1)


@pytest.mark.parametrize("channels,bitrate,frame", [(2, 128000, 960), (2, 64000, 480), (1, 32000, 960), (2, 6000, 960), (1, 500, 960)])
def test_emu_hard_cbr(channels, bitrate, frame):
    _run(channels, signals.music(8, channels=channels, seed=31), frame, 8 * 960 // frame, bitrate=bitrate, complexity=10, use_vbr=0)
