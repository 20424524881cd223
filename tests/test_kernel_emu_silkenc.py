"""CPU: the SILK encoder and the SILK-capable Opus layer (opus_amd/csrc/silk_enc*.h, opus_enc_sh.h — the exact device source) run on the 64-fiber wave
emulator and compared with the compiled reference: (a) below the Opus layer against silk_Encode (oracle/ref_expose/x_silk_enc.c: return code, byte
count, range-coder state, payload, control read-back; per-stage taps name the first diverging stage on failure), (b) whole packets against opus_encode
(SILK-only and hybrid: length, final range, bytes).  The checker is the compiled reference itself ("kind: reference"); there is no separate restatement."""
import ctypes, numpy as np, pytest
from reflib import ref_expose, ref_fx
pytestmark = pytest.mark.skipif(ref_expose() is None or not hasattr(ref_expose(), "refx_silk_encode"), reason="compiled reference (oracle/_ref) not built")

def signal(fs, secs, ch, seed):
    rng = np.random.default_rng(seed)
    t = np.arange(int(fs * secs)) / fs
    outs = []
    for c in range(ch):
        f0 = 120 + 30 * np.sin(2 * np.pi * 0.7 * t + c) + 15 * c
        ph = 2 * np.pi * np.cumsum(f0) / fs
        s = sum(np.sin(k * ph) / k for k in range(1, 25) if k * 150 < fs / 2) * (np.sin(2 * np.pi * 1.5 * t + 0.3 * c) > -0.3) * 6000 + rng.normal(0, 60 + 400 * (t > secs * 0.7), len(t))
        outs.append(s)
    return np.clip(np.stack(outs, 1).reshape(-1), -32768, 32767).astype(np.int16)

def run_silk(nframes, ch=1, chi=None, fs=16000, ms=20, seed=1, gain=1.0, **kw):
    from silkenc_harness import Pair, make_ctl, compare
    chi = chi or ch
    p = Pair(ch)
    pcm = (signal(fs, nframes * ms / 1000 + 0.1, ch, seed) * gain).astype(np.int16)
    n = fs * ms // 1000 * ch
    ctl = make_ctl(nChannelsAPI=ch, nChannelsInternal=chi, API_sampleRate=fs, payloadSize_ms=ms, **kw)
    for f in range(nframes):
        out, logs = p.step(ctl, pcm[f * n:(f + 1) * n].copy())
        d = compare(out, logs)
        assert d is None, (f, d)

@pytest.mark.parametrize("cx", [0, 1, 2, 5, 8, 10])
def test_emu_silk_encode_complexities(cx): run_silk(14, complexity=cx)
@pytest.mark.parametrize("gain", [0.0, 0.0004, 0.003, 0.02, 5.0])
def test_emu_silk_encode_levels(gain):
    """digital silence, a few LSBs (the low-level branches of the Burg recursion and the correlation scalings), and clipping input"""
    run_silk(14, gain=gain); run_silk(8, gain=gain, ch=2, bitRate=36000, complexity=4)
@pytest.mark.parametrize("kw", [
    dict(fs=24000), dict(fs=24000, desiredInternalSampleRate=12000, maxInternalSampleRate=12000, bitRate=16000), dict(fs=24000, desiredInternalSampleRate=8000, maxInternalSampleRate=8000, bitRate=12000),
    dict(fs=48000, desiredInternalSampleRate=12000, maxInternalSampleRate=12000, bitRate=16000), dict(fs=48000, ch=2, desiredInternalSampleRate=8000, maxInternalSampleRate=8000, bitRate=20000),
    dict(fs=12000, bitRate=16000), dict(fs=12000, desiredInternalSampleRate=8000, maxInternalSampleRate=8000, bitRate=12000), dict(fs=8000, bitRate=12000), dict(fs=48000, ms=10)])
def test_emu_silk_encode_resampler_ratios(kw):
    """every input-resampler method the encoder can select: copy, 1/2, 2/3, 3/4 (18-tap interpolated), 1/3, 1/4, 1/6 (symmetric 24 / 36 taps)"""
    run_silk(8, **dict(kw))
@pytest.mark.parametrize("kw", [
    dict(desiredInternalSampleRate=8000, maxInternalSampleRate=8000, bitRate=12000), dict(desiredInternalSampleRate=12000, maxInternalSampleRate=12000, bitRate=16000),
    dict(ms=10), dict(fs=48000), dict(useCBR=1, maxBits=60 * 8), dict(maxBits=40 * 8, bitRate=32000), dict(ch=2, bitRate=40000), dict(ch=2, fs=48000, bitRate=36000, complexity=5),
    dict(ch=2, chi=1), dict(ms=40), dict(ms=60, ch=2, bitRate=30000), dict(bitRate=6000, desiredInternalSampleRate=8000, maxInternalSampleRate=8000)])
def test_emu_silk_encode_matrix(kw):
    kw = dict(kw); nf = 8 if kw.get("ms", 20) > 20 else 14
    run_silk(nf, **kw)

CFG = ["Fs", "channels", "application", "user_bitrate_bps", "use_vbr", "vbr_constraint", "complexity", "force_channels", "user_bandwidth", "max_bandwidth", "lsb_depth", "disable_inv",
       "packet_loss_perc", "user_forced_mode", "signal_type", "use_inband_fec", "use_dtx"]
REQ = dict(user_bitrate_bps=4002, use_vbr=4006, vbr_constraint=4020, complexity=4010, force_channels=4022, user_bandwidth=4008, max_bandwidth=4004, user_forced_mode=11002, signal_type=4024, packet_loss_perc=4014, use_dtx=4016, use_inband_fec=4012)
def run_opus(nframes, Fs=16000, ch=1, app=2048, ms=20, seed=1, max_bytes=1276, shape=None, **kw):
    from silkenc_harness import build_emu, P
    E = build_emu(); R = ref_fx()
    R.opus_encoder_create.restype = ctypes.c_void_p; R.opus_encoder_ctl.argtypes = None; R.opus_encode.argtypes = None   # (other tests set prototypes on the shared handle)
    err = ctypes.c_int(0)
    enc = ctypes.c_void_p(R.opus_encoder_create(Fs, ch, app, ctypes.byref(err)))
    st = np.zeros(E.emu_sh_stream_size() + 64, np.uint8); E.emu_sh_stream_init(P(st), Fs, ch, app)
    for k, v in kw.items():
        assert R.opus_encoder_ctl(enc, REQ[k], ctypes.c_int(v)) == 0; E.emu_sh_set_cfg(P(st), CFG.index(k), v)
    n = int(Fs * ms) // 1000
    pcm = signal(Fs, nframes * ms / 1000 + 0.1, ch, seed)
    lens = []
    for f in range(nframes):
        x = pcm[f * n * ch:(f + 1) * n * ch].copy()
        if shape is not None: x = shape(f, x)
        o0 = np.zeros(1500, np.uint8); l0 = R.opus_encode(enc, P(x), n, P(o0), max_bytes)
        lens.append(l0)
        R.opus_encoder_ctl.argtypes = None; r0 = ctypes.c_uint32(0); R.opus_encoder_ctl(enc, 4031, ctypes.byref(r0))
        o1 = np.zeros(1500, np.uint8); l1 = np.zeros(1, np.int32); r1 = np.zeros(1, np.uint32)
        E.emu_sh_encode(P(st), P(x), n, max_bytes, P(o1), 1500, P(l1), P(r1))
        assert l0 == l1[0] and r0.value == r1[0] and np.array_equal(o0[:max(l0, 0)], o1[:max(l0, 0)]), (f, l0, int(l1[0]), r0.value, int(r1[0]))
    R.opus_encoder_destroy(enc)
    return lens

@pytest.mark.parametrize("kw", [
    dict(user_forced_mode=1000, user_bitrate_bps=24000, complexity=10),                                         # BASELINE config 3
    dict(user_bitrate_bps=16000), dict(app=2052, user_bitrate_bps=20000),
    dict(Fs=48000, app=2049, user_forced_mode=1000, user_bandwidth=1103, user_bitrate_bps=20000),
    dict(Fs=48000, ch=2, user_forced_mode=1000, user_bandwidth=1103, user_bitrate_bps=36000),
    dict(user_forced_mode=1000, use_vbr=0, user_bitrate_bps=16000), dict(Fs=8000, user_bitrate_bps=10000),
    dict(ms=60, user_forced_mode=1000, user_bitrate_bps=20000), dict(ms=10, user_forced_mode=1000, user_bitrate_bps=20000),
    dict(user_forced_mode=1000, user_bitrate_bps=32000, max_bytes=50)])
def test_emu_opus_silk_only(kw):
    kw = dict(kw); run_opus(8 if kw.get("ms", 20) > 20 else 14, **kw)

@pytest.mark.parametrize("kw", [
    dict(Fs=48000, ch=2, app=2049, user_forced_mode=1001, user_bandwidth=1105, user_bitrate_bps=128000, complexity=10),   # BASELINE config 4
    dict(Fs=48000, ch=1, app=2049, user_forced_mode=1001, user_bandwidth=1105, user_bitrate_bps=48000),
    dict(Fs=48000, ch=1, app=2048, user_forced_mode=1001, user_bandwidth=1104, user_bitrate_bps=32000),
    dict(Fs=48000, ch=2, app=2049, user_forced_mode=1001, user_bandwidth=1105, user_bitrate_bps=64000, use_vbr=0),
    dict(Fs=48000, ch=1, app=2049, ms=10, user_forced_mode=1001, user_bandwidth=1105, user_bitrate_bps=40000),
    dict(Fs=48000, ch=2, app=2049, user_forced_mode=1001, user_bandwidth=1104, user_bitrate_bps=24000),
    dict(Fs=48000, ch=1, app=2048, user_bitrate_bps=28000)])
def test_emu_opus_hybrid(kw): run_opus(12, **dict(kw))

@pytest.mark.parametrize("kw", [
    dict(Fs=48000, ch=1, app=2049),                                                                              # AUDIO default: the encoder's own decision is CELT-only
    dict(Fs=48000, ch=2, app=2049, user_forced_mode=1002, user_bitrate_bps=96000, complexity=10),
    dict(Fs=48000, ch=1, app=2048, ms=10, user_forced_mode=1002, user_bitrate_bps=32000),                       # VOIP: hp_cutoff in front of CELT
    dict(Fs=48000, ch=2, app=2049, ms=5, user_bitrate_bps=96000), dict(Fs=48000, ch=1, app=2049, ms=2.5, user_bitrate_bps=64000),
    dict(Fs=48000, ch=1, app=2049, user_forced_mode=1002, use_vbr=0, user_bitrate_bps=48000),
    dict(Fs=48000, ch=2, app=2048, user_forced_mode=1002, user_bitrate_bps=24000)])
def test_emu_opus_celt_only_in_audio_voip(kw):
    """CELT-only frames of an AUDIO / VOIP encoder: delay compensation (4 ms), the application's high-pass, the same CELT core"""
    run_opus(14, **dict(kw))

@pytest.mark.parametrize("kw", [
    dict(Fs=48000, ch=2, app=2049), dict(Fs=16000, ch=2, app=2048, user_bitrate_bps=24000), dict(Fs=48000, ch=2, app=2048, user_bitrate_bps=40000), dict(Fs=48000, ch=2, app=2049, user_bitrate_bps=36000)])
def test_emu_opus_stereo_automatic_mode(kw):
    """stereo input with the mode left to the encoder: compute_stereo_width feeds the SILK/CELT threshold"""
    run_opus(30, **dict(kw))


def _pause(lo, hi, level):
    """frames lo..hi-1 replaced by a +-level LSB dither (level 0: digital silence)"""
    def f(i, x):
        if lo <= i < hi: return ((np.arange(x.size) * 7919 % (2 * level + 1)) - level).astype(np.int16) if level else np.zeros_like(x)
        return x
    return f
@pytest.mark.parametrize("kw", [
    dict(user_forced_mode=1000, user_bitrate_bps=20000),                                   # SILK's own DTX on a noise floor; the generalised one on digital silence
    dict(Fs=48000, ch=2, app=2048, user_bitrate_bps=28000),
    dict(Fs=48000, ch=1, app=2049, user_forced_mode=1001, user_bandwidth=1105, user_bitrate_bps=40000),
    dict(Fs=48000, ch=1, app=2049, user_bitrate_bps=64000)])                                # CELT-only inside AUDIO: only digital silence goes quiet
def test_emu_opus_dtx(kw):
    """OPUS_SET_DTX: SILK DTX (no-speech counter, empty payload -> TOC-only packets, src/opus_encoder.c:2242) and the generalised decision on digital
    silence (:2565); packets must agree byte for byte including the 1-byte ones"""
    kw = dict(kw)
    for level in (2, 0):
        lens = run_opus(34, use_dtx=1, shape=_pause(4, 30, level), seed=3, **kw)
        if level == 0 or kw.get("user_forced_mode", 0) in (1000, 1001): assert 1 in lens[10:30], lens


@pytest.mark.parametrize("kw", [
    dict(user_forced_mode=1000, user_bitrate_bps=24000, packet_loss_perc=10),
    dict(user_bitrate_bps=20000, packet_loss_perc=20, complexity=10),
    dict(Fs=48000, ch=2, app=2048, user_bitrate_bps=40000, packet_loss_perc=15),
    dict(Fs=48000, ch=1, app=2048, user_bitrate_bps=32000, packet_loss_perc=8, user_bandwidth=1104, user_forced_mode=1001),
    dict(ms=40, user_forced_mode=1000, user_bitrate_bps=24000, packet_loss_perc=12), dict(ms=60, ch=2, user_forced_mode=1000, user_bitrate_bps=36000, packet_loss_perc=30),
    dict(user_forced_mode=1000, user_bitrate_bps=12000, packet_loss_perc=3), dict(user_forced_mode=1000, user_bitrate_bps=16000, packet_loss_perc=25, complexity=1)])
def test_emu_opus_inband_fec(kw):
    """OPUS_SET_INBAND_FEC + packet loss: decide_fec, the LBRR re-quantisation of every active frame and its coding at the head of the next packet"""
    kw = dict(kw); ms = kw.get("ms", 20)
    run_opus(16 if ms <= 20 else 8, use_inband_fec=1, seed=5, **kw)
