"""opus_projection_* (ambisonics with mixing / demixing matrices, mapping family 3: src/opus_projection_encoder.c, opus_projection_decoder.c, mapping_matrix.c) against
the compiled reference for every ambisonics order the reference has matrices for (1 to 5, with and without the two non-diegetic channels): stream layout, the demixing
matrix handed to the application (gain, size, bytes), packets and final range call by call (int16 and float entry points), and the projection decoder's PCM.
Here on the wave emulator; tests/test_gpu_surround.py runs it on the MI355X."""
import ctypes, numpy as np, pytest
import capi, signals
from reflib import ref_fx
pytestmark = pytest.mark.skipif(ref_fx() is None, reason="oracle/_ref not built")
WHICH = "emu"
vp, ci = ctypes.c_void_p, ctypes.c_int

def _enc(L, Fs, nch, app):
    L.opus_projection_ambisonics_encoder_create.restype = vp
    L.opus_projection_ambisonics_encoder_create.argtypes = [ci, ci, ci, ctypes.POINTER(ci), ctypes.POINTER(ci), ci, ctypes.POINTER(ci)]
    s, c, err = ci(), ci(), ci()
    e = L.opus_projection_ambisonics_encoder_create(Fs, nch, 3, ctypes.byref(s), ctypes.byref(c), app, ctypes.byref(err))
    L.opus_projection_encoder_destroy.argtypes = [vp]; L.opus_projection_encoder_destroy.restype = None
    return e, err.value, s.value, c.value

def _get(L, e, req):
    L.opus_projection_encoder_ctl.argtypes = [vp, ci, vp]; v = ctypes.c_int32()
    assert L.opus_projection_encoder_ctl(e, req, ctypes.byref(v)) == 0, req
    return v.value

def _matrix(L, e):
    size = _get(L, e, 6003); gain = _get(L, e, 6001)                     # OPUS_PROJECTION_GET_DEMIXING_MATRIX_SIZE / _GAIN
    buf = (ctypes.c_ubyte * size)()
    L.opus_projection_encoder_ctl.argtypes = [vp, ci, vp, ci]
    assert L.opus_projection_encoder_ctl(e, 6005, buf, size) == 0          # OPUS_PROJECTION_GET_DEMIXING_MATRIX
    assert L.opus_projection_encoder_ctl(e, 6005, buf, size - 1) == -1
    return size, gain, bytes(buf)

def _signal(nch, nsamp, seed):
    cols = [(signals.music(nsamp // 960 + 2, channels=1, seed=seed + c).reshape(-1)[:nsamp] * (0.9 / (1 + c % 5))).astype(np.int16) for c in range(nch)]
    return np.ascontiguousarray(np.stack(cols, 1))

@pytest.mark.parametrize("nch", [4, 6, 9, 11, 16, 18, 25, 27, 36, 38])
def test_projection_encoder_and_decoder(nch):
    R, E = capi.load("ref"), capi.load(WHICH)
    Fs, frame, nframes = 48000, 960, 6
    res = []
    sig = _signal(nch, frame * nframes, 3 * nch)
    for L in (R, E):
        e, err, s, c = _enc(L, Fs, nch, 2051)
        assert e and err == 0, (nch, err)
        L.opus_projection_encoder_ctl.argtypes = [vp, ci, ci]
        assert L.opus_projection_encoder_ctl(e, 4002, 48000 * (s + c)) == 0
        m = _matrix(L, e)
        L.opus_projection_encode.argtypes = [vp, vp, ci, vp, ci]
        buf = (ctypes.c_ubyte * 40000)(); pk = []
        for i in range(nframes):
            x = np.ascontiguousarray(sig[i * frame:(i + 1) * frame])
            n = L.opus_projection_encode(e, x.ctypes.data, frame, buf, 40000)
            pk.append((n, bytes(buf[:max(n, 0)]), _get(L, e, 4031) & 0xffffffff))
        L.opus_projection_encoder_destroy(e)
        res.append((s, c, m, pk))
    assert res[0][:3] == res[1][:3], (nch, res[0][:2], res[1][:2], res[0][2][:2], res[1][2][:2])
    assert res[0][3] == res[1][3], (nch, [a[0] for a in res[0][3]], [b[0] for b in res[1][3]])
    s, c, (size, gain, mat), pk = res[0]
    out = []
    for L in (R, E):
        L.opus_projection_decoder_create.restype = vp
        L.opus_projection_decoder_create.argtypes = [ci, ci, ci, ci, vp, ci, ctypes.POINTER(ci)]
        err = ci(); mb = (ctypes.c_ubyte * size).from_buffer_copy(mat)
        d = L.opus_projection_decoder_create(Fs, nch, s, c, mb, size, ctypes.byref(err))
        assert d and err.value == 0, (nch, err.value)
        L.opus_projection_decode.argtypes = [vp, ctypes.c_char_p, ci, vp, ci, ci]
        L.opus_projection_decoder_destroy.argtypes = [vp]; L.opus_projection_decoder_destroy.restype = None
        seq = []
        for n, p, _ in pk:
            o = np.zeros((frame, nch), np.int16)
            r = L.opus_projection_decode(d, p, n, o.ctypes.data, frame, 0)
            seq.append((r, o.tobytes()))
        L.opus_projection_decoder_destroy(d); out.append(seq)
    assert out[0] == out[1], nch

def test_projection_rejects_what_the_reference_rejects():
    R, E = capi.load("ref"), capi.load(WHICH)
    for nch in list(range(0, 40)) + [49, 51, 64, 66, 227, 255]:
        for fam in (0, 1, 2, 3, 255):
            a = _enc(R, 48000, nch, 2049) if fam == 3 else None
            if fam != 3: continue
            b = _enc(E, 48000, nch, 2049)
            assert (bool(a[0]), a[1]) == (bool(b[0]), b[1]) and (not a[0] or a[2:] == b[2:]), (nch, a[1:], b[1:])
            if a[0]: R.opus_projection_encoder_destroy(a[0]); E.opus_projection_encoder_destroy(b[0])

def test_projection_float_entry_points():
    R, E = capi.load("ref_fxa"), capi.load(WHICH)                         # the fixed-point reference that has the float entry points
    nch, frame = 9, 480
    sig = (_signal(nch, frame * 8, 5).astype(np.float32) / 32768.0)
    res = []
    for L in (R, E):
        e, err, s, c = _enc(L, 48000, nch, 2049); assert e and err == 0
        L.opus_projection_encoder_ctl.argtypes = [vp, ci, ci]
        assert L.opus_projection_encoder_ctl(e, 4010, 5) == 0                                   # below complexity 10: no tonality analysis on either side
        L.opus_projection_encode_float.argtypes = [vp, vp, ci, vp, ci]
        buf = (ctypes.c_ubyte * 20000)(); pk = []
        for i in range(8):
            x = np.ascontiguousarray(sig[i * frame:(i + 1) * frame])
            n = L.opus_projection_encode_float(e, x.ctypes.data, frame, buf, 20000); pk.append(bytes(buf[:max(n, 0)]) if n > 0 else n)
        L.opus_projection_encoder_destroy(e); res.append(pk)
    assert res[0] == res[1]

def test_projection_encode24_at_complexity_10():
    """opus_projection_encode24 with the tonality analysis on (complexity 10): the reference hands its int32 input to the analysis through downmix_int -- the int16 reader --
    and codes at MAX_ENCODING_DEPTH (src/opus_projection_encoder.c:408-415); a drop-in produces the same packets"""
    R, E = capi.load("ref_fxa"), capi.load(WHICH)
    nch, frame = 4, 960
    sig = _signal(nch, frame * 10, 11).astype(np.int32) << 8               # 24-bit samples
    sig[frame * 4:frame * 5] //= 4096                                        # a very quiet frame: the depth handed to CELT's dynalloc noise floor matters there
    res = []
    for L in (R, E):
        e, err, s, c = _enc(L, 48000, nch, 2049); assert e and err == 0
        L.opus_projection_encoder_ctl.argtypes = [vp, ci, ci]
        assert L.opus_projection_encoder_ctl(e, 4010, 10) == 0 and L.opus_projection_encoder_ctl(e, 4002, 4 * 48000) == 0
        if L is E: assert L.opus_projection_encoder_ctl(e, 11900, 1) == 0                     # (the test process starts encoders with the analysis off: conftest.py)
        L.opus_projection_encode24.argtypes = [vp, vp, ci, vp, ci]
        buf = (ctypes.c_ubyte * 20000)(); pk = []
        for i in range(10):
            x = np.ascontiguousarray(sig[i * frame:(i + 1) * frame])
            n = L.opus_projection_encode24(e, x.ctypes.data, frame, buf, 20000); pk.append(bytes(buf[:max(n, 0)]) if n > 0 else n)
        L.opus_projection_encoder_destroy(e); res.append(pk)
    assert all(isinstance(p, bytes) and len(p) > 8 for p in res[0])
    assert res[0] == res[1]
