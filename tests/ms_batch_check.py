"""tests/ms_batch_check.py — shared body of the device-resident multistream batch tests (opusgpu_ms_enc_batch_*, opus_amd/csrc/opus_ms_batch.h): B encoders of one
layout against B reference opus_multistream_encode calls on the same PCM, packet bytes and final range, over consecutive frames.  `which` = "emu" (CPU wave
emulator) or "gpu"."""
import ctypes, numpy as np
import capi
from test_kernel_emu_silkdec import speechy

def check(which, B, channels, streams, coupled, mapping, application, family=255, frames=4, frame=960, Fs=48000, bitrate=None, ctl=(), max_bytes=None):
    """max_bytes: the caller's buffer (a list: one value per frame, cycled) -- below the size at which every stream is offered the elementary encoder's own cap the reference
    chains the streams' byte budgets (opus_multistream_encoder.c:1016-1027) and so does the batch; error codes (a buffer too small for the packet) must agree as well"""
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
        mb = cap if max_bytes is None else int(max_bytes[f % len(max_bytes)])
        r = L.opusgpu_ms_encode_batch(m, pcm.ctypes.data, frame, out.ctypes.data, cap, mb, lens.ctypes.data, rng.ctypes.data)
        for b in range(B):
            n = R.opus_multistream_encode(refs[b], pcm[b].ctypes.data, frame, o.ctypes.data, mb)
            if r != 0: assert n == r, (f, b, n, r); continue                    # the whole call turned away (a buffer below the smallest packet): the reference says the same for every encoder
            fr = ctypes.c_uint32(); R.opus_multistream_encoder_ctl.argtypes = [vp, ci, vp]; R.opus_multistream_encoder_ctl(refs[b], 4031, ctypes.byref(fr))
            assert n == int(lens[b]), (f, b, mb, n, int(lens[b]))
            if n <= 0: continue
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
    # 64 elementary streams and more in one launch: the SILK-capable encoder's kernel pipeline against the reference
    dict(B=2, channels=40, streams=40, coupled=0, mapping=list(range(40)), application=2049, bitrate=40 * 64000, frames=3),              # config-5 shape, 80 mono AUDIO streams: CELT-only frames, kept by the front kernel and coded by the back kernel
    dict(B=2, channels=36, streams=36, coupled=0, mapping=list(range(36)), application=2048, Fs=16000, frame=320, bitrate=36 * 20000, frames=3),   # 72 VOIP streams at 16 kHz: SILK frames through front / quantiser / back
    dict(B=2, channels=48, streams=32, coupled=16, mapping=list(range(48)), application=2049, bitrate=32 * 48000, frames=3),             # 64 streams, half of them coupled pairs: hybrid / SILK / CELT as the rate split decides
]

# tight buffers: the streams' byte budgets chain (each stream is offered what its predecessors left)
TIGHT_CASES = [
    dict(B=3, channels=6, streams=4, coupled=2, mapping=[0, 1, 2, 3, 4, 5], application=2049, bitrate=256000, frames=8, max_bytes=[700, 400, 2000, 260, 120, 60, 9000, 30]),
    dict(B=2, channels=5, streams=5, coupled=0, mapping=[0, 1, 2, 3, 4], application=2049, bitrate=5 * 64000, frames=6, max_bytes=[500, 300, 200, 100, 40, 9]),       # ... down to the smallest legal packet (2 * streams - 1 bytes)
    dict(B=2, channels=4, streams=3, coupled=1, mapping=[0, 1, 2, 3], application=2051, bitrate=200000, frame=480, frames=6, max_bytes=[300, 150, 80, 1000, 20, 8]),
    dict(B=2, channels=3, streams=2, coupled=1, mapping=[0, 1, 2], application=2048, Fs=16000, frame=320, bitrate=60000, frames=6, max_bytes=[120, 60, 30, 200, 10, 4]),
    dict(B=2, channels=5, streams=3, coupled=1, mapping=[2, 0, 1, 255, 3], application=2049, bitrate=180000, frame=1920, frames=4, max_bytes=[900, 500, 250, 2500]),    # multi-frame elementary packets
    dict(B=2, channels=4, streams=4, coupled=0, mapping=[0, 1, 2, 3], application=2049, bitrate=4 * 96000, frame=4800, frames=3, max_bytes=[3000, 1200, 7]),         # 100 ms: one more byte kept back per stream
]

# hard CBR (OPUS_SET_VBR(0) on the multistream encoder): the packet is the bitrate's size, the last stream takes what the others leave and is padded out to it, padded elementary
# packets lose their padding on the way (opus_multistream_encoder.c:918-927, :1027, :1032-1048)
CBR_CASES = [
    dict(B=3, channels=6, streams=4, coupled=2, mapping=[0, 1, 2, 3, 4, 5], application=2049, bitrate=256000, frames=6, ctl=[(4006, 0)], max_bytes=[4000, 700, 400, 4000, 120, 30]),
    dict(B=2, channels=5, streams=5, coupled=0, mapping=[0, 1, 2, 3, 4], application=2049, bitrate=5 * 64000, frames=5, ctl=[(4006, 0)]),                        # the caller's buffer never binds: the bitrate does
    dict(B=2, channels=4, streams=3, coupled=1, mapping=[0, 1, 2, 3], application=2051, bitrate=200000, frame=480, frames=6, ctl=[(4006, 0)], max_bytes=[1000, 150, 80, 1000, 20, 8]),
    dict(B=2, channels=3, streams=2, coupled=1, mapping=[0, 1, 2], application=2048, Fs=16000, frame=320, bitrate=60000, frames=6, ctl=[(4006, 0)], max_bytes=[400, 60, 30, 200, 10, 4]),   # SILK elementary streams
    dict(B=2, channels=5, streams=3, coupled=1, mapping=[2, 0, 1, 255, 3], application=2049, bitrate=180000, frame=1920, frames=4, ctl=[(4006, 0)], max_bytes=[4000, 500, 250, 2500]),      # multi-frame elementary packets
    dict(B=2, channels=4, streams=4, coupled=0, mapping=[0, 1, 2, 3], application=2049, frames=4, ctl=[(4006, 0)]),                                                   # OPUS_AUTO: the streams' allocated rates set the size
    dict(B=2, channels=3, streams=3, coupled=0, mapping=[0, 1, 2], application=2049, bitrate=-1, frames=3, ctl=[(4006, 0)], max_bytes=[3000, 900, 200]),                # OPUS_BITRATE_MAX: only the buffer bounds it
    dict(B=2, channels=4, streams=4, coupled=0, mapping=[0, 1, 2, 3], application=2048, Fs=16000, frame=320, bitrate=4 * 9000, frames=8, ctl=[(4006, 0), (4016, 1)]),  # DTX under CBR: one- and two-byte elementary packets, padded
    dict(B=2, channels=4, streams=4, coupled=0, mapping=[0, 1, 2, 3], application=2049, bitrate=4 * 96000, frame=4800, frames=3, ctl=[(4006, 0)], max_bytes=[6000, 1200, 7]),   # 100 ms
]

# ---- projection (mapping family 3) encoder batch and the multistream / projection decoder batches ----
def _proj_ref_enc(R, Fs, channels, app):
    vp, ci = ctypes.c_void_p, ctypes.c_int
    R.opus_projection_ambisonics_encoder_create.restype = vp
    R.opus_projection_ambisonics_encoder_create.argtypes = [ci, ci, ci, ctypes.POINTER(ci), ctypes.POINTER(ci), ci, ctypes.POINTER(ci)]
    s, c, err = ci(), ci(), ci()
    e = R.opus_projection_ambisonics_encoder_create(Fs, channels, 3, ctypes.byref(s), ctypes.byref(c), app, ctypes.byref(err)); assert e and err.value == 0
    return e, s.value, c.value

def check_projection(which, B=2, channels=9, application=2049, frames=4, frame=960, Fs=48000, bitrate=None, complexity=None, analysis=False):
    """B projection encoders (device mixing in front of the multistream batch) against the reference's opus_projection_encode; then the packets through the projection DECODER
    batch (device parse / decode / demix) against the reference's opus_projection_decode"""
    L = capi.load(which); R = capi.load("ref_fxa" if analysis else "ref")
    vp, ci = ctypes.c_void_p, ctypes.c_int
    refs = [_proj_ref_enc(R, Fs, channels, application) for _ in range(B)]
    streams, coupled = refs[0][1], refs[0][2]
    L.opusgpu_ms_enc_batch_create.restype = vp
    L.opusgpu_ms_enc_batch_create.argtypes = [ci, ci, ci, ci, ci, ci, ctypes.c_char_p, ci, ci, ctypes.POINTER(ci)]
    L.opusgpu_ms_enc_batch_destroy.argtypes = [vp]; L.opusgpu_ms_enc_batch_destroy.restype = None
    L.opusgpu_ms_enc_batch_ctl.argtypes = [vp, ci, ci]
    L.opusgpu_ms_encode_batch.argtypes = [vp, vp, ci, vp, ci, ci, vp, vp]
    err = ci()
    m = L.opusgpu_ms_enc_batch_create(B, Fs, channels, 3, streams, coupled, bytes(range(channels)), application, 0, ctypes.byref(err)); assert m and err.value == 0, err.value
    sets = ([(4002, bitrate)] if bitrate else []) + ([(4010, complexity)] if complexity is not None else [])
    R.opus_projection_encoder_ctl.argtypes = [vp, ci, ci]
    for req, v in sets:
        assert L.opusgpu_ms_enc_batch_ctl(m, req, v) == 0, (req, v)
        for r in refs: assert R.opus_projection_encoder_ctl(r[0], req, v) == 0
    assert L.opusgpu_ms_enc_batch_ctl(m, 11900, 1 if analysis else 0) == 0
    # the decoder side: the demixing matrix from the reference encoder
    R.opus_projection_encoder_ctl.argtypes = [vp, ci, vp]; sz = ctypes.c_int32(); assert R.opus_projection_encoder_ctl(refs[0][0], 6003, ctypes.byref(sz)) == 0
    mat = (ctypes.c_ubyte * sz.value)(); R.opus_projection_encoder_ctl.argtypes = [vp, ci, vp, ci]; assert R.opus_projection_encoder_ctl(refs[0][0], 6005, mat, sz.value) == 0
    R.opus_projection_decoder_create.restype = vp; R.opus_projection_decoder_create.argtypes = [ci, ci, ci, ci, vp, ci, ctypes.POINTER(ci)]
    R.opus_projection_decode.argtypes = [vp, vp, ci, vp, ci, ci]
    rdec = [R.opus_projection_decoder_create(Fs, channels, streams, coupled, mat, sz.value, ctypes.byref(err)) for _ in range(B)]; assert all(rdec) and err.value == 0
    L.opusgpu_projection_dec_batch_create.restype = vp; L.opusgpu_projection_dec_batch_create.argtypes = [ci, ci, ci, ci, ci, vp, ci, ci, ctypes.POINTER(ci)]
    L.opusgpu_ms_decode_batch.argtypes = [vp, vp, ci, vp, vp, ci, vp, vp]
    L.opusgpu_ms_dec_batch_destroy.argtypes = [vp]; L.opusgpu_ms_dec_batch_destroy.restype = None
    d = L.opusgpu_projection_dec_batch_create(B, Fs, channels, streams, coupled, mat, sz.value, 0, ctypes.byref(err)); assert d and err.value == 0, err.value
    cap = (streams - 1) * 1279 + 7662 + 3 * streams + 8
    sig = [np.stack([(speechy(frames * frame // 960 + 2, 1, 17 * b + c, 960)[:, 0] * (0.8 / (1 + c % 4))).astype(np.int16) for c in range(channels)], 1) for b in range(B)]
    out = np.zeros((B, cap), np.uint8); lens = np.zeros(B, np.int32); rng = np.zeros(B, np.uint32)
    o = np.zeros(cap, np.uint8); R.opus_projection_encode.argtypes = [vp, vp, ci, vp, ci]
    pcm_out = np.zeros((B, frame, channels), np.int16); ns = np.zeros(B, np.int32); drng = np.zeros(B, np.uint32); ro = np.zeros((frame, channels), np.int16)
    for f in range(frames):
        pcm = np.ascontiguousarray(np.stack([x[f * frame:(f + 1) * frame] for x in sig]).astype(np.int16))
        r = L.opusgpu_ms_encode_batch(m, pcm.ctypes.data, frame, out.ctypes.data, cap, cap, lens.ctypes.data, rng.ctypes.data); assert r == 0, r
        for b in range(B):
            n = R.opus_projection_encode(refs[b][0], pcm[b].ctypes.data, frame, o.ctypes.data, cap)
            assert n == int(lens[b]) and bytes(out[b, :n]) == bytes(o[:n]), (f, b, n, int(lens[b]))
        if f == 2: lens[0] = 0                                                         # a lost packet for decoder 0: every stream conceals
        r = L.opusgpu_ms_decode_batch(d, out.ctypes.data, cap, lens.ctypes.data, pcm_out.ctypes.data, frame, ns.ctypes.data, drng.ctypes.data); assert r == 0, r
        for b in range(B):
            k = R.opus_projection_decode(rdec[b], out[b].ctypes.data if lens[b] else None, int(lens[b]), ro.ctypes.data, frame, 0)
            assert k == int(ns[b]) == frame, (f, b, k, int(ns[b]))
            assert np.array_equal(ro, pcm_out[b]), (f, b, int(np.abs(ro.astype(int) - pcm_out[b]).max()))
    L.opusgpu_ms_enc_batch_destroy(m); L.opusgpu_ms_dec_batch_destroy(d)
    R.opus_projection_encoder_destroy.argtypes = [vp]; R.opus_projection_decoder_destroy.argtypes = [vp]
    for r in refs: R.opus_projection_encoder_destroy(r[0])
    for r in rdec: R.opus_projection_decoder_destroy(r)

def check_ms_decode(which, B, channels, streams, coupled, mapping, application, frames=4, frame=960, Fs=48000, bitrate=None, corrupt=False):
    """packets of the reference's multistream encoder through the multistream DECODER batch against the reference's opus_multistream_decode: PCM, sample counts, final ranges;
    corrupt: one decoder gets a truncated packet (the reference's error code, its streams untouched) and is then fed good packets again"""
    L = capi.load(which); R = capi.load("ref")
    vp, ci = ctypes.c_void_p, ctypes.c_int
    R.opus_multistream_encoder_create.restype = vp; R.opus_multistream_encoder_create.argtypes = [ci, ci, ci, ci, ctypes.c_char_p, ci, ctypes.POINTER(ci)]
    R.opus_multistream_encode.argtypes = [vp, vp, ci, vp, ci]
    R.opus_multistream_decoder_create.restype = vp; R.opus_multistream_decoder_create.argtypes = [ci, ci, ci, ci, ctypes.c_char_p, ctypes.POINTER(ci)]
    R.opus_multistream_decode.argtypes = [vp, vp, ci, vp, ci, ci]
    L.opusgpu_ms_dec_batch_create.restype = vp; L.opusgpu_ms_dec_batch_create.argtypes = [ci, ci, ci, ci, ci, ctypes.c_char_p, ci, ctypes.POINTER(ci)]
    L.opusgpu_ms_decode_batch.argtypes = [vp, vp, ci, vp, vp, ci, vp, vp]
    L.opusgpu_ms_dec_batch_destroy.argtypes = [vp]; L.opusgpu_ms_dec_batch_destroy.restype = None
    err = ci(); mp = bytes(mapping)
    encs = [R.opus_multistream_encoder_create(Fs, channels, streams, coupled, mp, application, ctypes.byref(err)) for _ in range(B)]; assert all(encs)
    decs = [R.opus_multistream_decoder_create(Fs, channels, streams, coupled, mp, ctypes.byref(err)) for _ in range(B)]; assert all(decs)
    if bitrate:
        R.opus_multistream_encoder_ctl.argtypes = [vp, ci, ci]
        for e in encs: assert R.opus_multistream_encoder_ctl(e, 4002, bitrate) == 0
    d = L.opusgpu_ms_dec_batch_create(B, Fs, channels, streams, coupled, mp, 0, ctypes.byref(err)); assert d and err.value == 0, err.value
    nf = max(1, (frame * 50 + Fs - 1) // Fs); cap = (streams - 1) * (1276 * nf + 3) + 7662 + 3 * streams + 8
    step = 48000 // Fs
    sig = [np.ascontiguousarray(np.stack([speechy((frames * frame * step) // 960 + 2, 1, 41 * b + c, 960)[:, 0] for c in range(channels)], 1)[::step]) for b in range(B)]
    data = np.zeros((B, cap), np.uint8); lens = np.zeros(B, np.int32)
    pcm_out = np.zeros((B, frame, channels), np.int16); ns = np.zeros(B, np.int32); drng = np.zeros(B, np.uint32); ro = np.zeros((frame, channels), np.int16); fr = ctypes.c_uint32()
    R.opus_multistream_decoder_ctl.argtypes = [vp, ci, vp]
    for f in range(frames):
        for b in range(B):
            x = np.ascontiguousarray(sig[b][f * frame:(f + 1) * frame].astype(np.int16))
            n = R.opus_multistream_encode(encs[b], x.ctypes.data, frame, data[b].ctypes.data, cap); assert n > 0
            lens[b] = n
        if corrupt and f == 1: lens[B - 1] = max(2 * streams - 1, lens[B - 1] // 2)       # truncated in the middle of a stream
        r = L.opusgpu_ms_decode_batch(d, data.ctypes.data, cap, lens.ctypes.data, pcm_out.ctypes.data, frame, ns.ctypes.data, drng.ctypes.data); assert r == 0, r
        for b in range(B):
            k = R.opus_multistream_decode(decs[b], data[b].ctypes.data, int(lens[b]), ro.ctypes.data, frame, 0)
            assert k == int(ns[b]), (f, b, k, int(ns[b]))
            if k > 0:
                R.opus_multistream_decoder_ctl(decs[b], 4031, ctypes.byref(fr))
                assert np.array_equal(ro[:k], pcm_out[b, :k]) and fr.value == int(drng[b]), (f, b)
    L.opusgpu_ms_dec_batch_destroy(d)

DEC_CASES = [
    dict(B=3, channels=6, streams=4, coupled=2, mapping=[0, 4, 1, 2, 3, 5], application=2049, bitrate=256000, corrupt=True),
    dict(B=2, channels=5, streams=3, coupled=1, mapping=[2, 0, 1, 255, 3], application=2049, bitrate=180000, frame=1920, frames=3),      # 40 ms multi-frame elementary packets, a muted channel
    dict(B=2, channels=3, streams=2, coupled=1, mapping=[0, 1, 2], application=2048, Fs=16000, frame=320, bitrate=60000),                 # SILK streams at 16 kHz
]

def check_surround(which, B, channels, application=2049, frames=4, frame=960, Fs=48000, bitrate=None):
    """B surround encoders (mapping family 1: masking analysis + energy masks on the device) against the reference's opus_multistream_surround_encoder + opus_multistream_encode"""
    L = capi.load(which); R = capi.load("ref")
    vp, ci = ctypes.c_void_p, ctypes.c_int
    R.opus_multistream_surround_encoder_create.restype = vp
    R.opus_multistream_surround_encoder_create.argtypes = [ci, ci, ci, ctypes.POINTER(ci), ctypes.POINTER(ci), ctypes.c_char_p, ci, ctypes.POINTER(ci)]
    R.opus_multistream_encode.argtypes = [vp, vp, ci, vp, ci]
    err, s, c = ci(), ci(), ci(); refs = []
    for b in range(B):
        mp = ctypes.create_string_buffer(8)
        r = R.opus_multistream_surround_encoder_create(Fs, channels, 1, ctypes.byref(s), ctypes.byref(c), mp, application, ctypes.byref(err)); assert r and err.value == 0
        refs.append(r)
    streams, coupled = s.value, c.value
    L.opusgpu_ms_enc_batch_create.restype = vp
    L.opusgpu_ms_enc_batch_create.argtypes = [ci, ci, ci, ci, ci, ci, ctypes.c_char_p, ci, ci, ctypes.POINTER(ci)]
    L.opusgpu_ms_enc_batch_destroy.argtypes = [vp]; L.opusgpu_ms_enc_batch_destroy.restype = None
    L.opusgpu_ms_enc_batch_ctl.argtypes = [vp, ci, ci]
    L.opusgpu_ms_encode_batch.argtypes = [vp, vp, ci, vp, ci, ci, vp, vp]
    m = L.opusgpu_ms_enc_batch_create(B, Fs, channels, 1, streams, coupled, mp.raw, application, 0, ctypes.byref(err)); assert m and err.value == 0, err.value
    if bitrate:
        assert L.opusgpu_ms_enc_batch_ctl(m, 4002, bitrate) == 0
        R.opus_multistream_encoder_ctl.argtypes = [vp, ci, ci]
        for r in refs: assert R.opus_multistream_encoder_ctl(r, 4002, bitrate) == 0
    cap = (streams - 1) * 1279 + 7662 + 3 * streams + 8
    sig = [np.stack([(speechy(frames * frame // 960 + 2, 1, 23 * b + ch, 960)[:, 0] * (0.9 / (1 + ch % 3))).astype(np.int16) for ch in range(channels)], 1) for b in range(B)]
    out = np.zeros((B, cap), np.uint8); lens = np.zeros(B, np.int32); rng = np.zeros(B, np.uint32); o = np.zeros(cap, np.uint8)
    for f in range(frames):
        pcm = np.ascontiguousarray(np.stack([x[f * frame:(f + 1) * frame] for x in sig]).astype(np.int16))
        r = L.opusgpu_ms_encode_batch(m, pcm.ctypes.data, frame, out.ctypes.data, cap, cap, lens.ctypes.data, rng.ctypes.data); assert r == 0, r
        for b in range(B):
            n = R.opus_multistream_encode(refs[b], pcm[b].ctypes.data, frame, o.ctypes.data, cap)
            assert n == int(lens[b]) and bytes(out[b, :n]) == bytes(o[:n]), (f, b, n, int(lens[b]))
    L.opusgpu_ms_enc_batch_destroy(m)

def check_ms_decode_slot_limit(which):
    """an elementary packet beyond the decoder batch's 7,696-byte slot (seven 2.5 ms frames of 1,275 bytes, code 3) in the FIRST stream of a two-stream packet: the whole
    multistream packet is answered OPUS_BAD_ARG and neither elementary decoder has moved -- the good packets that follow decode exactly as they do in a batch that never saw it"""
    L = capi.load(which); R = capi.load("ref"); vp, ci = ctypes.c_void_p, ctypes.c_int
    R.opus_multistream_encoder_create.restype = vp; R.opus_multistream_encoder_create.argtypes = [ci, ci, ci, ci, ctypes.c_char_p, ci, ctypes.POINTER(ci)]
    R.opus_multistream_encode.argtypes = [vp, vp, ci, vp, ci]
    L.opusgpu_ms_dec_batch_create.restype = vp; L.opusgpu_ms_dec_batch_create.argtypes = [ci, ci, ci, ci, ci, ctypes.c_char_p, ci, ctypes.POINTER(ci)]
    L.opusgpu_ms_decode_batch.argtypes = [vp, vp, ci, vp, vp, ci, vp, vp]
    L.opusgpu_ms_dec_batch_destroy.argtypes = [vp]; L.opusgpu_ms_dec_batch_destroy.restype = None
    err = ci(); mp = bytes([0, 1])
    enc = R.opus_multistream_encoder_create(48000, 2, 2, 0, mp, 2051, ctypes.byref(err)); assert enc
    cap = 12000; frame = 960
    sig = np.stack([speechy(8, 1, 7 + c, 960)[:, 0] for c in range(2)], 1).astype(np.int16)
    good = []
    for f in range(5):
        buf = np.zeros(cap, np.uint8); x = np.ascontiguousarray(sig[f * frame:(f + 1) * frame])
        n = R.opus_multistream_encode(enc, x.ctypes.data, frame, buf.ctypes.data, cap); assert n > 0
        good.append((buf, n))
    rs = np.random.default_rng(3)
    toc = 0x80 | (0 << 3) | 3                                                    # CELT-only NB 2.5 ms, code 3 (mono)
    s0 = bytes([toc, 7, 252 + (1275 - 252) % 4, (1275 - (252 + (1275 - 252) % 4)) // 4]) + rs.integers(0, 256, 7 * 1275, dtype=np.uint8).tobytes()   # self-delimited: CBR, 7 frames of 1,275 bytes
    s1 = bytes([toc, 7]) + rs.integers(0, 256, 7 * 20, dtype=np.uint8).tobytes()
    bad = np.zeros(cap, np.uint8); bad[:len(s0) + len(s1)] = np.frombuffer(s0 + s1, np.uint8)
    def run(with_bad):
        d = L.opusgpu_ms_dec_batch_create(1, 48000, 2, 2, 0, mp, 0, ctypes.byref(err)); assert d and err.value == 0
        outs = []
        seq = [(good[0], False), (good[1], False)] + ([((bad, len(s0) + len(s1)), True)] if with_bad else []) + [(good[2], False), (good[3], False), (good[4], False)]
        for (buf, n), isbad in seq:
            pcm = np.zeros((1, frame, 2), np.int16); ns = np.zeros(1, np.int32); rg = np.zeros(1, np.uint32); lens = np.array([n], np.int32)
            r = L.opusgpu_ms_decode_batch(d, np.ascontiguousarray(buf).ctypes.data, cap, lens.ctypes.data, pcm.ctypes.data, frame, ns.ctypes.data, rg.ctypes.data); assert r == 0, r
            if isbad: assert int(ns[0]) == -1, int(ns[0])                       # OPUS_BAD_ARG for the whole packet
            else: outs.append((int(ns[0]), pcm.tobytes(), int(rg[0])))
        L.opusgpu_ms_dec_batch_destroy(d)
        return outs
    a, b = run(False), run(True)
    assert a == b and all(o[0] == frame for o in a)
