"""GPU (MI355X): the SILK-capable Opus encoder (applications VOIP / AUDIO / RESTRICTED_SILK, opus_amd/csrc/opus_enc_sh.h + silk_enc*.h) called
through the C ABI must produce the packets, lengths and final ranges of the compiled reference's opus_encode (oracle/_ref/libopus_ref_fx.so),
frame after frame with the state carried on the device.  BASELINE config 3 = 16 kHz mono VOIP, SILK-only, 20 ms, complexity 10."""
import ctypes, numpy as np, pytest
from reflib import ref_fx

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(ref_fx() is None, reason="compiled reference did not travel")]
PIPELINE = -1     # OPUS_AMD_SET_KERNEL_PIPELINE of the batches under test (include/opus_amd.h): tests/test_gpu_pipeline.py runs this module's matrix again with 1 = front / quantiser / back kernels
REQ = dict(bitrate=4002, vbr=4006, vbr_constraint=4020, complexity=4010, force_channels=4022, bandwidth=4008, max_bandwidth=4004, force_mode=11002, signal=4024, dtx=4016, fec=4012, loss=4014)

def speech(fs, secs, ch, seed):
    """SURVEY §8d config-3 signal: glottal-like harmonic source with vibrato, gated, over noise; one seed per stream"""
    rng = np.random.default_rng(seed)
    t = np.arange(int(fs * secs)) / fs
    outs = []
    for c in range(ch):
        f0 = 120 + 30 * np.sin(2 * np.pi * 0.7 * t + c + seed) + 15 * c + (seed % 7) * 9
        ph = 2 * np.pi * np.cumsum(f0) / fs
        s = sum(np.sin(k * ph) / k for k in range(1, 25) if k * 220 < fs / 2) * (np.sin(2 * np.pi * 1.5 * t + 0.3 * c + seed) > -0.3) * 6000 + rng.normal(0, 60 + 400 * (t > secs * 0.7), len(t))
        outs.append(s)
    return np.clip(np.stack(outs, 1), -32768, 32767).astype(np.int16)

class RefOpusEnc:
    def __init__(self, Fs, ch, app, **ctl):
        self.R = ref_fx(); self.R.opus_encoder_create.restype = ctypes.c_void_p; self.R.opus_encoder_ctl.argtypes = None; self.R.opus_encode.argtypes = None   # (other tests set prototypes on the shared handle)
        err = ctypes.c_int(0)
        self.enc = ctypes.c_void_p(self.R.opus_encoder_create(Fs, ch, app, ctypes.byref(err))); assert err.value == 0
        for k, v in ctl.items(): assert self.R.opus_encoder_ctl(self.enc, REQ[k], ctypes.c_int(v)) == 0
    def encode(self, x, n, max_bytes=1276):
        o = np.zeros(1500, np.uint8)
        l = self.R.opus_encode(self.enc, x.ctypes.data_as(ctypes.c_void_p), n, o.ctypes.data_as(ctypes.c_void_p), max_bytes)
        self.R.opus_encoder_ctl.argtypes = None; self.R.opus_encode.argtypes = None; r = ctypes.c_uint32(0); self.R.opus_encoder_ctl(self.enc, 4031, ctypes.byref(r))
        return bytes(o[:max(l, 0)]), l, r.value
    def __del__(self):
        try: self.R.opus_encoder_destroy(self.enc)
        except Exception: pass

def check(S, frames, Fs=16000, ch=1, app=2048, ms=20, max_bytes=1276, shape=None, **ctl):
    import opus_amd as oa
    b = oa.EncoderBatch(S, channels=ch, application=app, Fs=Fs)
    b.ctl(11902, PIPELINE)
    for k, v in ctl.items(): b.ctl(REQ[k], v)
    refs = [RefOpusEnc(Fs, ch, app, **ctl) for _ in range(S)]
    n = int(Fs * ms) // 1000
    sigs = [speech(Fs, frames * ms / 1000 + 0.1, ch, 10 + s) for s in range(S)]
    if shape is not None: sigs = [shape(s, x, n) for s, x in enumerate(sigs)]
    all_lens = []
    for f in range(frames):
        pcm = np.stack([np.ascontiguousarray(sigs[s][f * n:(f + 1) * n]).reshape(-1) for s in range(S)])
        pk, lens, rng = b.encode(pcm, n, max_bytes)
        for s in range(S):
            a = refs[s].encode(np.ascontiguousarray(sigs[s][f * n:(f + 1) * n]), n, max_bytes)
            assert (a[0], a[1], a[2]) == (pk[s], int(lens[s]), int(rng[s])), (f, s, a[1], int(lens[s]), hex(a[2]), hex(int(rng[s])))
        all_lens.append([int(l) for l in lens])
        if ctl.get("dtx"):
            for s in range(S):
                v = ctypes.c_int(-1); refs[s].R.opus_encoder_ctl(refs[s].enc, 4049, ctypes.byref(v)); assert b.get(4049, s) == v.value, (f, s, v.value)     # OPUS_GET_IN_DTX
    b.close()
    return all_lens

def test_gpu_config3_silk_voip_16k():
    """BASELINE config 3: VOIP 16 kHz mono, forced SILK-only, wideband, 20 ms, complexity 10, 24 kb/s VBR; 24 streams x 50 frame-steps"""
    check(24, 50, force_mode=1000, bandwidth=1103, bitrate=24000, complexity=10)

@pytest.mark.parametrize("cx", [0, 1, 2, 4, 6, 8])
def test_gpu_silk_complexities(cx): check(4, 25, force_mode=1000, bitrate=20000, complexity=cx)

@pytest.mark.parametrize("Fs,ch,app,ms,ctl", [
    (16000, 1, 2048, 20, dict(bitrate=16000)),                                            # VOIP, automatic mode decision
    (16000, 1, 2052, 20, dict(bitrate=20000)),                                            # RESTRICTED_SILK
    (48000, 1, 2049, 20, dict(force_mode=1000, bandwidth=1103, bitrate=20000)),           # AUDIO at 48 kHz, 48 -> 16 kHz resampler
    (48000, 2, 2048, 20, dict(force_mode=1000, bandwidth=1103, bitrate=36000)),           # stereo
    (16000, 2, 2048, 20, dict(force_mode=1000, bitrate=30000, complexity=5)),
    (8000, 1, 2048, 20, dict(bitrate=10000)),                                             # narrowband
    (12000, 1, 2048, 20, dict(bitrate=12000)),                                            # mediumband API rate
    (24000, 1, 2048, 20, dict(force_mode=1000, bandwidth=1103, bitrate=18000)),
    (16000, 1, 2048, 10, dict(force_mode=1000, bitrate=20000)),
    (16000, 1, 2048, 40, dict(force_mode=1000, bitrate=20000)),
    (16000, 2, 2048, 60, dict(force_mode=1000, bitrate=32000)),
    (16000, 1, 2048, 20, dict(force_mode=1000, vbr=0, bitrate=16000)),                    # hard CBR (rate loop + padding)
    (16000, 1, 2048, 20, dict(force_mode=1000, bandwidth=1101, bitrate=9000)),            # WB input coded NB
    (16000, 1, 2048, 20, dict(force_mode=1000, bandwidth=1102, bitrate=12000)),
])
def test_gpu_silk_matrix(Fs, ch, app, ms, ctl): check(4, 25 if ms <= 20 else 10, Fs=Fs, ch=ch, app=app, ms=ms, **ctl)

@pytest.mark.parametrize("Fs,ch,app,ms,ctl", [
    (48000, 2, 2049, 20, dict(force_mode=1001, bandwidth=1105, bitrate=128000, complexity=10)),   # BASELINE config 4: hybrid audio 48 kHz stereo 20 ms VBR 128 kb/s
    (48000, 1, 2049, 20, dict(force_mode=1001, bandwidth=1105, bitrate=48000)),
    (48000, 1, 2048, 20, dict(force_mode=1001, bandwidth=1104, bitrate=32000)),                   # super-wideband
    (48000, 2, 2049, 20, dict(force_mode=1001, bandwidth=1105, bitrate=64000, vbr=0)),            # hybrid CBR
    (48000, 1, 2049, 10, dict(force_mode=1001, bandwidth=1105, bitrate=40000)),
    (48000, 2, 2049, 20, dict(force_mode=1001, bandwidth=1104, bitrate=24000)),
    (48000, 1, 2048, 20, dict(bitrate=28000)),                                                    # hybrid by the encoder's own mode / bandwidth decision
    (48000, 2, 2049, 20, dict(force_mode=1001, bandwidth=1105, bitrate=96000, complexity=3)),
])
def test_gpu_hybrid_matrix(Fs, ch, app, ms, ctl): check(4, 25, Fs=Fs, ch=ch, app=app, ms=ms, **ctl)

def test_gpu_config4_hybrid_long():
    """BASELINE config 4 shape over 60 frame-steps, 16 streams"""
    check(16, 60, Fs=48000, ch=2, app=2049, force_mode=1001, bandwidth=1105, bitrate=128000, complexity=10)

def test_gpu_silk_small_buffer():
    """max_data_bytes below the VBR demand: the rate-control loop (gain search, pulses cleared as a last resort) must take the reference's path"""
    check(4, 25, max_bytes=40, force_mode=1000, bitrate=32000)
    check(2, 15, max_bytes=25, force_mode=1000, bitrate=24000)

@pytest.mark.parametrize("Fs,ch,app,ms,ctl", [
    (48000, 1, 2049, 20, dict()),                                                                 # AUDIO default: CELT-only by the encoder's own decision
    (48000, 2, 2049, 20, dict()),                                                                 # stereo, automatic mode (compute_stereo_width)
    (48000, 2, 2049, 20, dict(force_mode=1002, bitrate=96000, complexity=10)),
    (48000, 1, 2048, 10, dict(force_mode=1002, bitrate=32000)),
    (48000, 2, 2049, 5, dict(bitrate=96000)), (48000, 1, 2049, 2.5, dict(bitrate=64000)),
    (48000, 2, 2048, 20, dict(bitrate=40000)), (16000, 2, 2048, 20, dict(bitrate=24000)),
])
def test_gpu_celt_only_and_auto_modes_in_audio_voip(Fs, ch, app, ms, ctl): check(4, 25, Fs=Fs, ch=ch, app=app, ms=ms, **ctl)

@pytest.mark.parametrize("Fs,ch,app,ctl", [
    (16000, 1, 2048, dict(force_mode=1000, bitrate=20000)), (48000, 2, 2048, dict(bitrate=28000)),
    (48000, 1, 2049, dict(force_mode=1001, bandwidth=1105, bitrate=40000)), (48000, 2, 2049, dict(bitrate=96000))])
def test_gpu_dtx(Fs, ch, app, ctl):
    """OPUS_SET_DTX: a noise-floor pause (SILK's own DTX) and a digital-silence pause (the generalised decision); TOC-only packets and OPUS_GET_IN_DTX must agree"""
    def shape(s, x, n):
        x = x.copy()
        x[4 * n:26 * n] = ((np.arange(22 * n * ch).reshape(-1, ch) * 7919 % 5) - 2) if s % 2 == 0 else 0
        return x
    lens = check(4, 30, Fs=Fs, ch=ch, app=app, shape=shape, dtx=1, **ctl)
    assert any(l[1] == 1 for l in lens[10:26])                                   # the digital-silence streams do go quiet

@pytest.mark.parametrize("Fs,ch,app,ms,ctl", [
    (16000, 1, 2048, 20, dict(force_mode=1000, bitrate=24000, loss=10, complexity=10)), (16000, 1, 2048, 20, dict(bitrate=26000, loss=20)),
    (48000, 2, 2048, 20, dict(bitrate=40000, loss=15)), (48000, 1, 2048, 20, dict(force_mode=1001, bandwidth=1104, bitrate=32000, loss=8)),
    (16000, 1, 2048, 40, dict(force_mode=1000, bitrate=24000, loss=12)), (16000, 2, 2048, 60, dict(force_mode=1000, bitrate=36000, loss=30)),
    (48000, 2, 2049, 20, dict(force_mode=1001, bandwidth=1105, bitrate=96000, loss=10, complexity=10))])
def test_gpu_inband_fec(Fs, ch, app, ms, ctl):
    """OPUS_SET_INBAND_FEC with packet loss: decide_fec, the LBRR re-quantisation (silk_LBRR_encode_FIX) and its coding at the head of the next packet"""
    check(4, 20 if ms <= 20 else 8, Fs=Fs, ch=ch, app=app, ms=ms, fec=1, **ctl)

def test_gpu_formerly_unbuilt_paths_match_the_reference():
    """round 1 refused CELT below 48 kHz and a SILK -> CELT switch mid-stream; both are built now and have to match the reference packet for packet"""
    import opus_amd as oa
    rs = np.random.default_rng(1)
    x = rs.normal(0, 3000, (12, 480)).astype(np.int16)
    b = oa.EncoderBatch(1, channels=1, application=2049, Fs=24000)              # AUDIO 24 kHz at the default rate decides CELT-only
    ref = RefOpusEnc(24000, 1, 2049)
    for f in range(12):
        pk, lens, rng = b.encode(x[f:f + 1], 480)
        assert bytes(pk[0]) == ref.encode(x[f], 480)[0], f
    b.close()
    b = oa.EncoderBatch(1, channels=1, application=2048, Fs=48000)              # SILK -> CELT-only switch: redundancy frame
    ref = RefOpusEnc(48000, 1, 2048, force_mode=1000, bitrate=20000)
    b.ctl(11002, 1000); b.ctl(4002, 20000)
    y = rs.normal(0, 3000, (8, 960)).astype(np.int16)
    for f in range(8):
        if f == 4: b.ctl(11002, 1002); assert ref.R.opus_encoder_ctl(ref.enc, 11002, ctypes.c_int(1002)) == 0
        pk, lens, rng = b.encode(y[f:f + 1], 960)
        assert bytes(pk[0]) == ref.encode(y[f], 960)[0], f
    b.close()
    with pytest.raises(oa.OpusError): oa.EncoderBatch(1, channels=1, application=2048, Fs=44100)

def test_gpu_classic_api_silk_memcpy_contract():
    """classic opus_encoder_* entry points for a VOIP encoder: the state blob is flat (the reference's tests memcpy it, tests/test_opus_encode.c:398-404)"""
    import opus_amd as oa
    L = oa.lib()
    e = oa.OpusEncoder(16000, 1, 2048); e.ctl(11002, 1000); e.ctl(4002, 24000); e.ctl(4010, 10)
    ref = RefOpusEnc(16000, 1, 2048, force_mode=1000, bitrate=24000, complexity=10)
    sig = speech(16000, 1.0, 1, 3)
    size = L.opus_encoder_get_size(1)
    for f in range(20):
        x = np.ascontiguousarray(sig[f * 320:(f + 1) * 320])
        if f == 8:                                                               # move the encoder to fresh memory, poison and free the old blob
            libc = ctypes.CDLL(None); libc.malloc.restype = ctypes.c_void_p; libc.malloc.argtypes = [ctypes.c_size_t]
            newp = libc.malloc(size); ctypes.memmove(newp, e._st, size); ctypes.memset(e._st, 0xFF, size)
            L.opus_encoder_destroy(ctypes.c_void_p(e._st)); e._st = newp
        pk = e.encode(x, 320)
        a = ref.encode(x, 320)
        assert pk == a[0] and e.final_range() == a[2], f

def test_gpu_silk_large_batch_replicas():
    """size-independent property at bench scale: 8 distinct signals tiled over 8,192 waves give identical results on every replica, 3 frame-steps"""
    import opus_amd as oa
    S = 8192
    b = oa.EncoderBatch(S, channels=1, application=2048, Fs=16000)
    for k, v in dict(force_mode=1000, bandwidth=1103, bitrate=24000, complexity=10).items(): b.ctl(REQ[k], v)
    refs = [RefOpusEnc(16000, 1, 2048, force_mode=1000, bandwidth=1103, bitrate=24000, complexity=10) for _ in range(8)]
    sigs = [speech(16000, 0.2, 1, 40 + s) for s in range(8)]
    for f in range(3):
        pcm = np.stack([sigs[s % 8][f * 320:(f + 1) * 320].reshape(-1) for s in range(S)])
        pk, lens, rng = b.encode(pcm, 320)
        for s8 in range(8):
            a = refs[s8].encode(np.ascontiguousarray(sigs[s8][f * 320:(f + 1) * 320]), 320)
            idx = np.arange(s8, S, 8)
            assert np.all(lens[idx] == a[1]) and np.all(rng[idx] == a[2]) and pk[s8] == a[0] and pk[S - 8 + s8] == a[0]
    b.close()
