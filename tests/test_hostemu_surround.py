"""The surround layouts (mapping family 1 with more than two channels: src/opus_multistream_encoder.c) against the compiled reference: the masking analysis on its own
(surround_analysis :230 -- the reference exports it -- against opusgpu_surround_analysis, signal-to-mask ratios and analysis memory word for word) and whole
multistream encoders (packets and final ranges call by call: masks, the surround rate allocation, the LFE stream, forced stereo CELT on coupled streams, bandwidth by
equivalent rate).  The reference's own test of these layouts (tests/opus_encode_regressions.c: surround_analysis_uninit) only checks that nothing crashes; a call-by-
call comparison of the two libraries under that program (tools/encode_trace_shim.c) is what showed that logSum() returns 16 bits in the fixed-point build.
Here on the wave emulator; tests/test_gpu_multistream.py runs the same tests on the MI355X."""
import ctypes, numpy as np, pytest
import capi, signals
from reflib import ref_fx, ref_fxa
from test_kernel_emu_silkdec import speechy
pytestmark = pytest.mark.skipif(ref_fx() is None or ref_fxa() is None, reason="oracle/_ref not built")
WHICH = "emu"
CB = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p)

def _copy_in(dst, dst_stride, src, src_stride, src_ch, n, user):           # opus_copy_channel_in_short (:757; static in the reference): opus_res = int16 in this build
    d = (ctypes.c_int16 * (n * dst_stride)).from_address(dst); s = (ctypes.c_int16 * (n * src_stride)).from_address(src)
    for i in range(n): d[i * dst_stride] = s[i * src_stride + src_ch]
_cb = CB(_copy_in)

def _signal(nch, nsamp, Fs, seed):
    cols = []
    for c in range(nch):
        x = signals.music(nsamp * (48000 // Fs) // 960 + 2, channels=1, seed=seed + c).reshape(-1) if c % 3 != 1 else speechy(nsamp * (48000 // Fs) // 960 + 2, 1, seed + c, 960).reshape(-1)
        g = (1.0, 0.5, 0.8, 0.05, 0.3, 1.0, 0.6, 0.2)[c]
        cols.append((x[::48000 // Fs][:nsamp] * g).astype(np.int16))
    return np.ascontiguousarray(np.stack(cols, 1))

@pytest.mark.parametrize("nch", [3, 4, 5, 6, 7, 8])
def test_surround_analysis_against_the_reference_function(nch):
    R = capi.load("ref"); E = capi.load(WHICH)
    R.opus_custom_mode_create.restype = ctypes.c_void_p
    mode = R.opus_custom_mode_create(48000, 960, None)
    R.surround_analysis.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 4 + [CB, ctypes.c_int]; R.surround_analysis.restype = None
    E.opusgpu_surround_analysis.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int32] + [ctypes.c_void_p] * 3
    for Fs, sizes in ((48000, (960, 480, 120, 240, 960)), (24000, (480, 480, 60)), (16000, (320, 160)), (12000, (240,)), (8000, (160, 20))):
        sig = _signal(nch, sum(sizes), Fs, 11 * nch); pos = 0
        if Fs == 16000: sig[100:200] = 32767; sig[200:260] = -32768
        m1 = np.zeros((nch, 120), np.int32); p1 = np.zeros(nch, np.int32); m2 = m1.copy(); p2 = p1.copy()
        for n in sizes:
            x = np.ascontiguousarray(sig[pos:pos + n]).reshape(-1); pos += n
            a = np.zeros((nch, 21), np.int32); b = np.zeros((nch, 21), np.int32)
            R.surround_analysis(mode, x.ctypes.data, a.ctypes.data, m1.ctypes.data, p1.ctypes.data, n, 120, nch, Fs, _cb, 0)
            assert E.opusgpu_surround_analysis(x.ctypes.data, n, nch, Fs, m2.ctypes.data, p2.ctypes.data, b.ctypes.data) == 0
            assert np.array_equal(a, b) and np.array_equal(m1, m2) and np.array_equal(p1, p2), (Fs, n, np.argwhere(a != b)[:4])
    assert E.opusgpu_surround_analysis(x.ctypes.data, n, 2, Fs, m2.ctypes.data, p2.ctypes.data, b.ctypes.data) == -1

def _ms(L, Fs, nch, family, app):
    L = capi._proto(L); streams = ctypes.c_int(); coupled = ctypes.c_int(); mapping = (ctypes.c_ubyte * 8)(); err = ctypes.c_int()
    L.opus_multistream_surround_encoder_create.restype = ctypes.c_void_p
    L.opus_multistream_surround_encoder_create.argtypes = [ctypes.c_int32, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    enc = L.opus_multistream_surround_encoder_create(Fs, nch, family, ctypes.byref(streams), ctypes.byref(coupled), mapping, app, ctypes.byref(err))
    assert enc and err.value == 0
    L.opus_multistream_encode.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int32]
    L.opus_multistream_encoder_destroy.argtypes = [ctypes.c_void_p]; L.opus_multistream_encoder_destroy.restype = None
    return enc, (streams.value, coupled.value, bytes(mapping[:nch]))

def _set(L, enc, req, v):
    L.opus_multistream_encoder_ctl.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    return L.opus_multistream_encoder_ctl(enc, req, v)
def _rng(L, enc):
    v = ctypes.c_uint32(); L.opus_multistream_encoder_ctl.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    assert L.opus_multistream_encoder_ctl(enc, 4031, ctypes.byref(v)) == 0
    return v.value

def run(ref, Fs, nch, app, sizes, ctls=(), sched=None, seed=1, maxb=12000, analysis=0):
    """the same surround encoder on the reference (`ref`: "ref" = no float API, "ref_fxa" = with the analysis) and on the library under test"""
    R = capi.load(ref); E = capi.load(WHICH)
    er, lr = _ms(R, Fs, nch, 1, app); ee, le = _ms(E, Fs, nch, 1, app)
    assert lr == le
    assert _set(E, ee, 11900, analysis) == 0
    for req, v in ctls: assert _set(R, er, req, v) == _set(E, ee, req, v) == 0, (req, v)
    sig = _signal(nch, sum(sizes), Fs, seed); pos = 0
    bufr = (ctypes.c_ubyte * maxb)(); bufe = (ctypes.c_ubyte * maxb)()
    lens = []
    for i, n in enumerate(sizes):
        for req, v in (sched or {}).get(i, ()): assert _set(R, er, req, v) == _set(E, ee, req, v) == 0, (req, v)
        x = np.ascontiguousarray(sig[pos:pos + n]).reshape(-1); pos += n
        a = R.opus_multistream_encode(er, x.ctypes.data, n, bufr, maxb); b = E.opus_multistream_encode(ee, x.ctypes.data, n, bufe, maxb)
        assert a == b and bytes(bufr[:max(a, 0)]) == bytes(bufe[:max(b, 0)]), (i, n, a, b)
        if a > 0: assert _rng(R, er) == _rng(E, ee), i
        lens.append(a)
    for L, e in ((R, er), (E, ee)): L.opus_multistream_encoder_destroy(e)
    return lens

@pytest.mark.parametrize("nch", [3, 4, 5, 6, 7, 8])
def test_surround_encoders_every_layout(nch):
    run("ref", 48000, nch, 2049, [960] * 12, seed=nch)
    run("ref", 48000, nch, 2049, [960] * 6 + [480] * 4 + [1920, 2880], ctls=((4002, 64000 * nch),), seed=20 + nch)

def test_surround_rates_applications_and_settings():
    run("ref", 24000, 3, 2049, [960, 1440, 480, 480], ctls=((4024, 3001), (4006, 1), (4020, 1), (4010, 0), (4004, 1101), (4008, 1101), (4036, 8), (4012, 1), (4002, 84315)))   # the regression's settings
    run("ref", 16000, 6, 2048, [320] * 10, ctls=((4002, 96000),), seed=3)                       # VOIP: uncoupled streams may leave CELT
    run("ref", 48000, 6, 2051, [480] * 10, ctls=((4002, 256000),), seed=4)
    run("ref", 48000, 6, 2049, [960] * 10, ctls=((4002, 96000), (4006, 0)), seed=5)             # hard CBR: stream by stream
    run("ref", 48000, 5, 2049, [960] * 16, sched={4: ((4002, 40000),), 8: ((4010, 3),), 12: ((4002, 400000), (4020, 0))}, seed=6)
    run("ref", 8000, 4, 2049, [160] * 8, seed=7)
    run("ref", 12000, 8, 2049, [240] * 6, ctls=((4002, -1),), seed=8)                            # OPUS_BITRATE_MAX

def test_surround_with_the_float_api_analysis():
    """complexity 10, API rate >= 16 kHz: the elementary encoders run the tonality analysis next to the masks"""
    run("ref_fxa", 48000, 6, 2049, [960] * 30, ctls=((4002, 192000),), seed=9, analysis=1)
    run("ref_fxa", 48000, 3, 2049, [960] * 20 + [1920] * 4, ctls=((4002, 60000),), seed=10, analysis=1)
    run("ref_fxa", 24000, 4, 2048, [480] * 20, seed=11, analysis=1)
