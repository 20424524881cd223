"""tests/ms_batch_check.py — shared body of the device-resident multistream batch tests (opusgpu_ms_enc_batch_*, opus_amd/csrc/opus_ms_batch.h): B encoders of one
layout against B reference opus_multistream_encode calls on the same PCM, packet bytes and final range, over consecutive frames.  `which` = "emu" (CPU wave
emulator) or "gpu"."""
import ctypes, numpy as np
import capi
from test_kernel_emu_silkdec import speechy

def check(which, B, channels, streams, coupled, mapping, application, family=255, frames=4, frame=960, Fs=48000, bitrate=None, ctl=()):
    L = capi.load(which); R = capi.load("ref")
    vp, ci = ctypes.c_void_p, ctypes.c_int
    L.opusgpu_ms_enc_batch_create.restype = vp
    L.opusgpu_ms_enc_batch_create.argtypes = [ci, ci, ci, ci, ci, ci, ctypes.c_char_p, ci, ci, ctypes.POINTER(ci)]
    L.opusgpu_ms_enc_batch_destroy.argtypes = [vp]; L.opusgpu_ms_enc_batch_destroy.restype = None
    L.opusgpu_ms_enc_batch_ctl.argtypes = [vp, ci, ci]
    L.opusgpu_ms_encode_batch.argtypes = [vp, vp, ci, vp, ci, ci, vp, vp]
    R.opus_multistream_encoder_create.restype = vp
    R.opus_multistream_encoder_create.argtypes = [ci, ci, ci, ci, ctypes.c_char_p, ci, ctypes.POINTER(ci)]
    R.opus_multistream_encode.argtypes = [vp, vp, ci, vp, ci]
    R.opus_multistream_encoder_destroy.argtypes = [vp]
    err = ci()
    mp = bytes(mapping)
    m = L.opusgpu_ms_enc_batch_create(B, Fs, channels, family, streams, coupled, mp, application, 0, ctypes.byref(err))
    assert m and err.value == 0, err.value
    refs = []
    for b in range(B):
        r = R.opus_multistream_encoder_create(Fs, channels, streams, coupled, mp, application, ctypes.byref(err)); assert r and err.value == 0
        refs.append(r)
    sets = list(ctl) + ([(4002, bitrate)] if bitrate else [])
    for req, v in sets:
        assert L.opusgpu_ms_enc_batch_ctl(m, req, v) == 0, (req, v)
        R.opus_multistream_encoder_ctl.argtypes = [vp, ci, ci]
        for r in refs: assert R.opus_multistream_encoder_ctl(r, req, v) == 0
    nf = max(1, (frame * 50 + Fs - 1) // Fs)
    cap = (streams - 1) * (1276 * nf + 3) + 7662 + 3 * streams + 8
    step = 48000 // Fs
    sig = [np.stack([speechy((frames * frame * step) // 960 + 2, 1, 31 * b + c, 960)[:, 0] for c in range(channels)], 1) for b in range(B)]          # [B] x [n, channels] at 48 kHz
    sig = [np.ascontiguousarray(x[::step]) for x in sig]
    out = np.zeros((B, cap), np.uint8); lens = np.zeros(B, np.int32); rng = np.zeros(B, np.uint32)
    o = np.zeros(cap, np.uint8)
    for f in range(frames):
        pcm = np.ascontiguousarray(np.stack([x[f * frame:(f + 1) * frame] for x in sig]).astype(np.int16))
        assert pcm.shape == (B, frame, channels)
        r = L.opusgpu_ms_encode_batch(m, pcm.ctypes.data, frame, out.ctypes.data, cap, cap, lens.ctypes.data, rng.ctypes.data)
        assert r == 0, r
        for b in range(B):
            n = R.opus_multistream_encode(refs[b], pcm[b].ctypes.data, frame, o.ctypes.data, cap)
            fr = ctypes.c_uint32(); R.opus_multistream_encoder_ctl.argtypes = [vp, ci, vp]; R.opus_multistream_encoder_ctl(refs[b], 4031, ctypes.byref(fr))
            assert n == int(lens[b]), (f, b, n, int(lens[b]))
            assert bytes(out[b, :n]) == bytes(o[:n]), (f, b, [k for k in range(n) if out[b, k] != o[k]][:6])
            assert fr.value == int(rng[b]), (f, b)
    L.opusgpu_ms_enc_batch_destroy(m)
    for r in refs: R.opus_multistream_encoder_destroy(r)

CASES = [
    dict(B=3, channels=6, streams=4, coupled=2, mapping=[0, 1, 2, 3, 4, 5], application=2049, bitrate=256000),              # AUDIO: 2 coupled + 2 mono
    dict(B=2, channels=5, streams=5, coupled=0, mapping=[0, 1, 2, 3, 4], application=2049, bitrate=5 * 64000),               # config-5 shape: all mono AUDIO
    dict(B=2, channels=4, streams=3, coupled=1, mapping=[0, 1, 2, 3], application=2051, bitrate=200000, frame=480),         # restricted-lowdelay, 10 ms
    dict(B=2, channels=3, streams=2, coupled=1, mapping=[0, 1, 2], application=2048, Fs=16000, frame=320, bitrate=60000),    # VOIP 16 kHz (SILK elementary streams)
    dict(B=2, channels=5, streams=3, coupled=1, mapping=[2, 0, 1, 255, 3], application=2049, bitrate=180000, frame=1920, frames=3),   # 40 ms calls (multi-frame elementary packets), a muted channel, permuted mapping
]
