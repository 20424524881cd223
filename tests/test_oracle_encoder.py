"""CPU: whole-encoder parity of the plain-C restatement (oracle/) against the compiled reference (fixed-point,
DISABLE_FLOAT_API) — identical packets and OPUS_GET_FINAL_RANGE, frame after frame, with state carried."""
import ctypes, numpy as np, pytest
from reflib import ref_fx, oracle
import signals

pytestmark = pytest.mark.skipif(ref_fx() is None or oracle() is None, reason="oracle/_ref or oracle lib not built")

class RefEnc:
    def __init__(self, channels, application=2051, **ctl):
        L = self.L = ref_fx()
        L.opus_encoder_create.restype = ctypes.c_void_p
        L.opus_encoder_create.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
        L.opus_encode.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
        L.opus_encoder_ctl.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        err = ctypes.c_int()
        self.st = L.opus_encoder_create(48000, channels, application, ctypes.byref(err))
        assert err.value == 0
        req = dict(bitrate=4002, complexity=4010, vbr=4006, vbr_constraint=4020, force_channels=4022, bandwidth=4008, max_bandwidth=4004, lsb_depth=4036, phase_inv_disabled=4046, force_mode=11002, signal=4024, inband_fec=4012, packet_loss=4014, dtx=4016)
        for k, v in ctl.items(): assert L.opus_encoder_ctl(self.st, req[k], v) == 0
        self.out = (ctypes.c_ubyte * 1500)()
    def encode(self, pcm, frame, maxb=1276):
        n = self.L.opus_encode(self.st, pcm.ctypes.data, frame, self.out, maxb)
        rng = ctypes.c_uint32()
        self.L.opus_encoder_ctl.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        self.L.opus_encoder_ctl(self.st, 4031, ctypes.byref(rng))
        self.L.opus_encoder_ctl.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        return bytes(self.out[:max(n, 0)]), n, rng.value

class OracleEnc:
    def __init__(self, channels, application=2051, **ctl):
        O = self.O = oracle()
        self.buf = ctypes.create_string_buffer(O.oc_opus_enc_size())
        assert O.oc_opus_enc_init(self.buf, 48000, channels, application) == 0
        what = dict(bitrate=0, complexity=1, vbr=2, use_vbr=2, vbr_constraint=3, force_channels=4, bandwidth=5, user_bandwidth=5, max_bandwidth=6, lsb_depth=7, phase_inv_disabled=8, disable_inv=8)
        for k, v in ctl.items(): assert O.oc_opus_enc_set(self.buf, what[k], v) == 0
        self.out = (ctypes.c_ubyte * 1500)()
        O.oc_opus_enc_final_range.restype = ctypes.c_uint32
        O.oc_opus_encode.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    def encode(self, pcm, frame, maxb=1276):
        n = self.O.oc_opus_encode(self.buf, pcm.ctypes.data, frame, self.out, maxb)
        return bytes(self.out[:max(n, 0)]), n, self.O.oc_opus_enc_final_range(self.buf)

def _run(channels, sig, frame, nframes, **ctl):
    r = RefEnc(channels, **ctl); o = OracleEnc(channels, **ctl)
    for i in range(nframes):
        pcm = np.ascontiguousarray(sig[i * frame:(i + 1) * frame])
        a = r.encode(pcm, frame); b = o.encode(pcm, frame)
        assert a == b, (i, a[1], b[1], hex(a[2]), hex(b[2]))

@pytest.mark.parametrize("seed", range(4))
def test_config2_stereo_128k_c10(seed):
    """BASELINE config 2: restricted-lowdelay 48 kHz stereo 20 ms, 128 kb/s CVBR, complexity 10."""
    _run(2, signals.music(150, seed=seed), 960, 150, bitrate=128000, complexity=10)

@pytest.mark.parametrize("channels,bitrate,complexity,frame", [
    (2, 64000, 10, 960), (2, 24000, 10, 960), (2, 256000, 10, 960), (2, 510000, 10, 960),
    (1, 64000, 10, 960), (1, 12000, 5, 960), (2, 96000, 5, 960), (2, 96000, 0, 960), (2, 48000, 3, 960), (1, 32000, 8, 960),
    (2, 128000, 10, 480), (2, 128000, 10, 240), (2, 128000, 10, 120), (1, 48000, 10, 480), (1, 64000, 7, 120), (2, 16000, 10, 960), (2, 8000, 10, 960)])
def test_rates_sizes(channels, bitrate, complexity, frame):
    n = 100 * 960 // frame
    _run(channels, signals.music(100, channels=channels, seed=7)[:], frame, min(n, 300), bitrate=bitrate, complexity=complexity)

@pytest.mark.parametrize("kind", ["bursts", "tone", "silence", "loud"])
def test_signal_kinds(kind):
    sig = dict(bursts=signals.noise_bursts(120), tone=signals.tone(120, freq=997.0), silence=signals.silence_then_music(120),
               loud=(signals.music(120, amp=60000.0)))[kind]
    _run(2, sig, 960, 120, bitrate=128000, complexity=10)

def test_unconstrained_vbr_and_ctls():
    _run(2, signals.music(80, seed=3), 960, 80, bitrate=96000, complexity=10, vbr_constraint=0)
    _run(2, signals.music(80, seed=4), 960, 80, bitrate=96000, complexity=10, force_channels=1)
    _run(2, signals.music(80, seed=5), 960, 80, bitrate=64000, complexity=10, bandwidth=1103)
    _run(2, signals.music(80, seed=6), 960, 80, bitrate=64000, complexity=10, max_bandwidth=1104, phase_inv_disabled=1)


@pytest.mark.parametrize("channels,bitrate,frame,maxb", [(2, 128000, 960, 1276), (2, 64000, 480, 1276), (1, 32000, 960, 1276), (2, 510000, 960, 1276),
                                                         (2, 96000, 960, 200), (2, 6000, 960, 1276), (2, 256000, 120, 1276), (1, 500, 960, 1276)])
def test_hard_cbr(channels, bitrate, frame, maxb):
    """OPUS_SET_VBR(0): cbr_bytes budget (opus_encoder.c:1328), CELT CBR allocation, code-3 padding of short / TOC-only packets"""
    sig = signals.music(12, channels=channels, seed=31)
    r = RefEnc(channels, bitrate=bitrate, complexity=10, vbr=0); o = OracleEnc(channels, bitrate=bitrate, complexity=10, vbr=0)
    sizes = set()
    for i in range(12 * 960 // frame):
        pcm = np.ascontiguousarray(sig[i * frame:(i + 1) * frame])
        a = r.encode(pcm, frame, maxb); b = o.encode(pcm, frame, maxb)
        assert a == b, (i, a[1], b[1], hex(a[2]), hex(b[2]))
        sizes.add(a[1])
    assert len(sizes) == 1            # constant bitrate: every packet has the same size
