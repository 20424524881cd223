"""GPU (MI355X): opus_multistream_* through the C ABI against the compiled reference (same entry points of oracle/_ref/libopus_ref_fx.so,
application RESTRICTED_LOWDELAY): identical multistream packets, OPUS_GET_FINAL_RANGE (xor over streams), decoded PCM.  Covers explicit
layouts (coupled + mono, duplicated and muted output channels), mapping families 0 / 2 / 255 incl. 255 channels in one launch, generous
buffers (all streams of a group in one launch), tight buffers and hard CBR (streams stepped in order)."""
import ctypes, numpy as np, pytest
import signals
from reflib import ref_fx

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(ref_fx() is None, reason="compiled reference did not travel")]

def _libs():
    import opus_amd
    A = opus_amd.lib(); R = ref_fx()
    vp = ctypes.c_void_p
    for L in (A, R):
        L.opus_multistream_encoder_create.restype = vp
        L.opus_multistream_encoder_create.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
        L.opus_multistream_surround_encoder_create.restype = vp
        L.opus_multistream_surround_encoder_create.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), vp, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
        L.opus_multistream_encode.argtypes = [vp, vp, ctypes.c_int, vp, ctypes.c_int]
        L.opus_multistream_encoder_destroy.argtypes = [vp]
        L.opus_multistream_decoder_create.restype = vp
        L.opus_multistream_decoder_create.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int)]
        L.opus_multistream_decode.argtypes = [vp, ctypes.c_char_p, ctypes.c_int, vp, ctypes.c_int, ctypes.c_int]
        L.opus_multistream_decoder_destroy.argtypes = [vp]
    return A, R

def _ctl_set(L, st, req, v, enc=True):
    f = L.opus_multistream_encoder_ctl if enc else L.opus_multistream_decoder_ctl
    f.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    return f(st, req, v)
def _ctl_get_u32(L, st, req, enc=True):
    f = L.opus_multistream_encoder_ctl if enc else L.opus_multistream_decoder_ctl
    f.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    v = ctypes.c_uint32()
    assert f(st, req, ctypes.byref(v)) == 0
    return v.value

def _sig(channels, frames, seed):
    cols = [signals.music(frames, channels=1, seed=seed + c).reshape(-1) if c % 3 else signals.noise_bursts(frames, channels=1, seed=seed + c).reshape(-1) for c in range(channels)]
    return np.ascontiguousarray(np.stack(cols, axis=1))

def _run(mk_enc, dec_layout, channels, frame, nframes, maxb, ctls=(), seed=1):
    A, R = _libs()
    encs = [mk_enc(L) for L in (A, R)]
    for L, e in zip((A, R), encs):
        for req, v in ctls: assert _ctl_set(L, e, req, v) == 0
    streams, coupled, mapping, out_ch = dec_layout
    decs = []
    for L in (A, R):
        err = ctypes.c_int()
        d = L.opus_multistream_decoder_create(48000, out_ch, streams, coupled, bytes(mapping), ctypes.byref(err))
        assert d and err.value == 0
        decs.append(d)
    sig = _sig(channels, nframes * frame // 960 + 1, seed)
    tocs = set()
    for i in range(nframes):
        pcm = np.ascontiguousarray(sig[i * frame:(i + 1) * frame])
        outs = []
        for L, e in zip((A, R), encs):
            buf = (ctypes.c_ubyte * maxb)()
            n = L.opus_multistream_encode(e, pcm.ctypes.data, frame, buf, maxb)
            outs.append((n, bytes(buf[:max(n, 0)]), _ctl_get_u32(L, e, 4031) if n > 0 else None))
        assert outs[0] == outs[1], (i, outs[0][0], outs[1][0], outs[0][2], outs[1][2])
        if outs[0][0] <= 0: continue
        pkt = outs[1][1]
        tocs.add(pkt[0] >> 3)
        dres = []
        for L, d in zip((A, R), decs):
            o = np.full((frame, out_ch), 7, np.int16)
            n = L.opus_multistream_decode(d, pkt, len(pkt), o.ctypes.data, frame, 0)
            dres.append((n, o.copy(), _ctl_get_u32(L, d, 4031, enc=False)))
        assert dres[0][0] == dres[1][0] == frame and dres[0][2] == dres[1][2] == outs[0][2], (i, dres[0][0], dres[1][0])
        assert np.array_equal(dres[0][1], dres[1][1]), (i, np.argwhere(dres[0][1] != dres[1][1])[:4])
    for L, e, d in zip((A, R), encs, decs): L.opus_multistream_encoder_destroy(e); L.opus_multistream_decoder_destroy(d)
    return tocs                                                                      # TOC configurations of the first elementary stream

def _explicit(channels, streams, coupled, mapping, app=2051):
    def mk(L):
        err = ctypes.c_int()
        e = L.opus_multistream_encoder_create(48000, channels, streams, coupled, bytes(mapping), app, ctypes.byref(err))
        assert e and err.value == 0, err.value
        return e
    return mk

def _family(channels, family, app=2051):
    info = {}
    def mk(L):
        err = ctypes.c_int(); s = ctypes.c_int(); c = ctypes.c_int(); m = (ctypes.c_ubyte * 256)()
        e = L.opus_multistream_surround_encoder_create(48000, channels, family, ctypes.byref(s), ctypes.byref(c), m, app, ctypes.byref(err))
        assert e and err.value == 0, err.value
        got = (s.value, c.value, list(m[:channels]))
        assert info.setdefault("layout", got) == got
        return e
    return mk, info

def test_ms_explicit_layout_generous_buffer():
    mapping = [0, 1, 2, 3, 4, 5]                      # 2 coupled streams (ch 0-3) + 2 mono
    _run(_explicit(6, 4, 2, mapping), (4, 2, mapping, 6), 6, 960, 12, 40000, ctls=[(4002, 256000), (4010, 10)])

def test_ms_decoder_layout_with_muted_and_duplicated_outputs():
    enc_map = [0, 1, 2]                                # 1 coupled + 1 mono
    dec_map = [2, 255, 0, 1, 0]                        # mono first, a muted channel, then L R L
    _run(_explicit(3, 2, 1, enc_map), (2, 1, dec_map, 5), 3, 480, 16, 20000, ctls=[(4002, 160000)])

@pytest.mark.parametrize("channels,family", [(1, 0), (2, 0), (2, 1), (4, 2), (6, 2), (11, 2), (5, 255), (255, 255)])
def test_ms_mapping_families(channels, family):
    mk, info = _family(channels, family)
    A, R = _libs()
    probe = mk(R); R.opus_multistream_encoder_destroy(probe)
    s, c, m = info["layout"]
    _run(mk, (s, c, m, channels), channels, 960, 4 if channels > 64 else 8, 255 * 1300 + 9000)

@pytest.mark.parametrize("channels,family,app", [(255, 255, 2049), (6, 255, 2049), (9, 2, 2049), (2, 1, 2049), (3, 255, 2048)])
def test_ms_general_applications(channels, family, app):
    """BASELINE config 5 as specified (255 mono streams, OPUS_APPLICATION_AUDIO) and the other non-restricted applications: the elementary encoders are
    the SILK-capable records, every stream makes the reference's own mode decision (CELT-only at these rates; ambisonics forces it)"""
    mk, info = _family(channels, family, app)
    A, R = _libs()
    probe = mk(R); R.opus_multistream_encoder_destroy(probe)
    s, c, m = info["layout"]
    _run(mk, (s, c, m, channels), channels, 960, 4 if channels > 64 else 8, 255 * 1300 + 9000)

def test_ms_voip_low_rate_streams_are_silk():
    """two mono + one coupled stream at speech rates: the elementary encoders choose SILK-only / hybrid"""
    mapping = [0, 1, 2, 3]
    t1 = _run(_explicit(4, 3, 1, mapping, app=2048), (3, 1, mapping, 4), 4, 960, 10, 40000, ctls=[(4002, 72000), (4010, 6)])
    t2 = _run(_explicit(2, 2, 0, [0, 1], app=2049), (2, 0, [0, 1], 2), 2, 960, 10, 40000, ctls=[(4002, 24000), (4024, 3001)])      # AUDIO, signal hint voice
    assert min(t1) < 16 and min(t2) < 16, (t1, t2)                                    # configurations 0-11 SILK-only, 12-15 hybrid

def test_ms_tight_buffer_and_cbr_go_stream_by_stream():
    mapping = [0, 1, 2, 3]
    _run(_explicit(4, 3, 1, mapping), (3, 1, mapping, 4), 4, 960, 10, 2000, ctls=[(4002, 510000)])                 # VBR capped by the caller's buffer
    _run(_explicit(4, 3, 1, mapping), (3, 1, mapping, 4), 4, 960, 10, 4000, ctls=[(4002, 192000), (4006, 0)])     # hard CBR: last stream absorbs the remainder, padded
    _run(_explicit(4, 3, 1, mapping), (3, 1, mapping, 4), 4, 240, 20, 4000, ctls=[(4002, 300000), (4006, 0)])

def test_ms_errors():
    A, R = _libs()
    for L in (A, R):
        err = ctypes.c_int()
        assert not L.opus_multistream_encoder_create(48000, 2, 1, 1, bytes([0, 0]), 2051, ctypes.byref(err)) and err.value == -1      # right channel unmapped
        assert not L.opus_multistream_encoder_create(48000, 300, 1, 0, bytes([0] * 300), 2051, ctypes.byref(err)) and err.value == -1
        assert not L.opus_multistream_decoder_create(48000, 2, 1, 0, bytes([0, 5]), ctypes.byref(err)) and err.value == -1
        e = L.opus_multistream_encoder_create(48000, 2, 2, 0, bytes([0, 1]), 2051, ctypes.byref(err))
        buf = (ctypes.c_ubyte * 10)()
        pcm = np.zeros((960, 2), np.int16)
        assert L.opus_multistream_encode(e, pcm.ctypes.data, 960, buf, 2) == -2                                                          # below the smallest packet
        assert L.opus_multistream_encode(e, pcm.ctypes.data, 100, buf, 10) == -1
        L.opus_multistream_encoder_destroy(e)
        d = L.opus_multistream_decoder_create(48000, 2, 2, 0, bytes([0, 1]), ctypes.byref(err))
        o = np.zeros((960, 2), np.int16)
        assert L.opus_multistream_decode(d, b"\xf8", 1, o.ctypes.data, 960, 0) == -4                                                     # shorter than 2*streams-1
        assert L.opus_multistream_decode(d, b"\xf8\x01\x00\xf0\x00", 5, o.ctypes.data, 960, 0) == -4                                    # streams with different durations
        L.opus_multistream_decoder_destroy(d)

def test_gpu_multistream_decode_silk_and_hybrid_streams():
    """a multistream packet whose elementary streams are SILK-only / hybrid (voip application of the reference encoder): the multistream decoder runs the
    same kernels per stream group, so every packet mode decodes"""
    import ctypes, opus_amd
    from reflib import ref_fx
    from test_kernel_emu_silkdec import speechy
    R = ref_fx(); L = opus_amd.lib()
    ch, streams, coupled = 3, 2, 1
    mapping = (ctypes.c_ubyte * 3)(0, 1, 2)
    err = ctypes.c_int()
    R.opus_multistream_encoder_create.restype = ctypes.c_void_p; R.opus_multistream_decoder_create.restype = ctypes.c_void_p
    L.opus_multistream_decoder_create.restype = ctypes.c_void_p
    R.opus_multistream_encoder_create.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    for Lx in (R, L): Lx.opus_multistream_decoder_create.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
    enc = R.opus_multistream_encoder_create(48000, ch, streams, coupled, mapping, 2048, ctypes.byref(err)); assert err.value == 0
    R.opus_multistream_encoder_ctl.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    assert R.opus_multistream_encoder_ctl(enc, 4002, 60000) == 0
    rd = R.opus_multistream_decoder_create(48000, ch, streams, coupled, mapping, ctypes.byref(err)); assert err.value == 0
    gd = L.opus_multistream_decoder_create(48000, ch, streams, coupled, mapping, ctypes.byref(err)); assert err.value == 0 and gd
    for Lx in (R, L): Lx.opus_multistream_decode.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    R.opus_multistream_encode.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    sig = np.concatenate([speechy(12, 2, 5), speechy(12, 1, 6)], axis=1).astype(np.int16)
    out = (ctypes.c_ubyte * 4000)(); modes = set()
    for i in range(12):
        n = R.opus_multistream_encode(enc, np.ascontiguousarray(sig[i * 960:(i + 1) * 960]).ctypes.data, 960, out, 4000)
        assert n > 0
        pkt = bytes(out[:n]); modes.add(pkt[0] >> 7)
        a = np.zeros((960, ch), np.int16); b = np.zeros((960, ch), np.int16)
        na = R.opus_multistream_decode(rd, pkt, n, a.ctypes.data, 960, 0); nb = L.opus_multistream_decode(gd, pkt, n, b.ctypes.data, 960, 0)
        assert na == nb == 960, (i, na, nb)
        assert np.array_equal(a, b), i
    assert 0 in modes                       # SILK or hybrid TOCs were really in there
