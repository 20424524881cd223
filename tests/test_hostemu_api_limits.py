"""CPU: corners of the classic API where the byte budget, not the signal, decides the result -- each against the compiled reference, call by call:
hard CBR above 510 kb/s (the packet is padded past the 1275 bytes a coded frame can fill, src/opus_encoder.c:1330,:2646), OPUS_BITRATE_MAX with a buffer larger than
six frames (multi-frame calls are padded to the caller's WHOLE buffer, :1757), opus_packet_pad to tens of kilobytes (hundreds of padding length bytes), a
multistream encoder that follows another one of the same shape but a different application, OPUS_SET_GAIN on a multistream decoder, a multistream sub-packet
longer than six maximum frames, and the frame_size check of opus_multistream_decode with decode_fec (opus_multistream_decoder.c:216-222).
tests/test_gpu_classic_api.py re-runs them on the MI355X against opus_amd/libopus_amd.so."""
import ctypes, numpy as np, pytest
import capi, signals
from reflib import ref_fx
from test_hostemu_encoder_modes import sig_for
pytestmark = pytest.mark.skipif(ref_fx() is None, reason="oracle/_ref not built")
WHICH = "emu"
vp, ci = ctypes.c_void_p, ctypes.c_int

def _enc_pair(Fs, ch, app, **ctl): return capi.Enc("ref", Fs, ch, app, **ctl), capi.Enc(WHICH, Fs, ch, app, **ctl)

def _encode_big(e, pcm, frame, maxb):
    out = (ctypes.c_ubyte * (maxb + 64))()
    pcm = np.ascontiguousarray(pcm, np.int16)
    n = e.L.opus_encode(e.st, pcm.ctypes.data, frame, out, maxb)
    return n, bytes(out[:max(n, 0)]), e.get(4031) & 0xffffffff

@pytest.mark.parametrize("app", [2049, 2051])
@pytest.mark.parametrize("ms", [20, 40, 60])
def test_hard_cbr_above_510_kbps(app, ms):
    Fs, ch = 48000, 2
    fr = Fs * ms // 1000
    for bitrate, maxb in ((700000, 6000), (-1, 8000), (1200000, 7000), (-1, 1500)):
        a, b = _enc_pair(Fs, ch, app, vbr=0, bitrate=bitrate)
        sig = sig_for(Fs, ch, fr * 3 + 16, 40 + ms)
        for i in range(3):
            x = _encode_big(a, sig[i * fr:(i + 1) * fr], fr, maxb); y = _encode_big(b, sig[i * fr:(i + 1) * fr], fr, maxb)
            assert x[0] == y[0], (app, ms, bitrate, maxb, i, x[0], y[0])
            assert x[0] > 1275 or maxb <= 1500
            assert x[2] == y[2] and x[1] == y[1], (app, ms, bitrate, maxb, i)

def test_bitrate_max_pads_a_long_call_to_the_whole_buffer():
    """60 ms, hard CBR, OPUS_BITRATE_MAX, 14,000-byte buffer: a code-3 packet of exactly 14,000 bytes whose padding length run (39 bytes) is longer than the
    staging head-room of the device-side assembly"""
    Fs, ch, fr, maxb = 48000, 1, 2880, 14000
    a, b = _enc_pair(Fs, ch, 2051, vbr=0, bitrate=-1)
    sig = sig_for(Fs, ch, fr * 2 + 16, 77)
    for i in range(2):
        x = _encode_big(a, sig[i * fr:(i + 1) * fr], fr, maxb); y = _encode_big(b, sig[i * fr:(i + 1) * fr], fr, maxb)
        assert x[0] == y[0] == maxb and x[1] == y[1] and x[2] == y[2]

def test_packet_pad_to_tens_of_kilobytes():
    R, E = capi.load("ref"), capi.load(WHICH)
    a = capi.Enc("ref", 48000, 2, 2049, bitrate=96000)
    sig = sig_for(48000, 2, 960 * 3, 3)
    pkt = a.encode(sig[960:1920], 960)[0]
    for new_len in (len(pkt) + 1, len(pkt) + 300, 10000, 60000, 200000):
        res = []
        for L in (R, E):
            buf = (ctypes.c_ubyte * new_len)(*pkt)
            L.opus_packet_pad.argtypes = [vp, ci, ci]; L.opus_packet_unpad.argtypes = [vp, ci]
            r = L.opus_packet_pad(buf, len(pkt), new_len)
            padded = bytes(buf)
            n = L.opus_packet_unpad(buf, new_len)
            res.append((r, padded, n, bytes(buf[:max(n, 0)])))
        assert res[0] == res[1], new_len
        assert res[1][0] == 0 and res[1][2] == len(pkt) and res[1][3] == pkt
    # multistream: the padding lands on the last stream
    for L in (R, E):
        L.opus_multistream_packet_pad.argtypes = [vp, ci, ci, ci]; L.opus_multistream_packet_unpad.argtypes = [vp, ci, ci]
    ms = _ms_packet(R, 2, 960)
    out = []
    for L in (R, E):
        buf = (ctypes.c_ubyte * 40000)(*ms)
        r = L.opus_multistream_packet_pad(buf, len(ms), 40000, 2)
        padded = bytes(buf)
        n = L.opus_multistream_packet_unpad(buf, 40000, 2)
        out.append((r, padded, n, bytes(buf[:max(n, 0)])))
    assert out[0] == out[1] and out[1][0] == 0 and out[1][3] == ms

def _ms_proto(L):
    L.opus_multistream_encoder_create.restype = vp
    L.opus_multistream_encoder_create.argtypes = [ci, ci, ci, ci, ctypes.c_char_p, ci, ctypes.POINTER(ci)]
    L.opus_multistream_encode.argtypes = [vp, vp, ci, vp, ci]
    L.opus_multistream_encoder_destroy.argtypes = [vp]; L.opus_multistream_encoder_destroy.restype = None
    L.opus_multistream_decoder_create.restype = vp
    L.opus_multistream_decoder_create.argtypes = [ci, ci, ci, ci, ctypes.c_char_p, ctypes.POINTER(ci)]
    L.opus_multistream_decode.argtypes = [vp, ctypes.c_char_p, ci, vp, ci, ci]
    L.opus_multistream_decoder_destroy.argtypes = [vp]; L.opus_multistream_decoder_destroy.restype = None
    return L

def _ms_enc(L, Fs, channels, streams, coupled, app):
    err = ci()
    e = _ms_proto(L).opus_multistream_encoder_create(Fs, channels, streams, coupled, bytes(range(channels)), app, ctypes.byref(err))
    assert e and err.value == 0, err.value
    return e

def _ms_set(L, st, req, v, enc=True):
    f = L.opus_multistream_encoder_ctl if enc else L.opus_multistream_decoder_ctl
    f.argtypes = [vp, ci, ci]
    return f(st, req, int(v))

def _ms_packet(L, channels, frame, Fs=48000, app=2051, seed=9, bitrate=None, vbr=None, maxb=4000):
    e = _ms_enc(L, Fs, channels, channels, 0, app)
    if bitrate is not None: assert _ms_set(L, e, 4002, bitrate) == 0
    if vbr is not None: assert _ms_set(L, e, 4006, vbr) == 0
    sig = np.stack([sig_for(Fs, 1, frame * 2, seed + c)[:, 0] for c in range(channels)], axis=1)
    buf = (ctypes.c_ubyte * maxb)()
    pcm = np.ascontiguousarray(sig[frame:2 * frame], np.int16)
    n = L.opus_multistream_encode(e, pcm.ctypes.data, frame, buf, maxb)
    L.opus_multistream_encoder_destroy(e)
    assert n > 0, n
    return bytes(buf[:n])

def test_ms_encoder_batch_follows_the_encoder_in_use():
    """a RESTRICTED_SILK multistream encoder, then an AUDIO one of the same shape coding 5 ms frames (which RESTRICTED_SILK refuses): the shared device batch
    must validate against the encoder in use"""
    R, E = capi.load("ref"), capi.load(WHICH)
    Fs = 48000
    res = []
    for L in (R, E):
        e1 = _ms_enc(L, Fs, 2, 2, 0, 2052); e2 = _ms_enc(L, Fs, 2, 2, 0, 2049)
        sig = np.stack([sig_for(Fs, 1, 4000, 5 + c)[:, 0] for c in range(2)], axis=1).astype(np.int16)
        buf = (ctypes.c_ubyte * 4000)()
        seq = []
        p = np.ascontiguousarray(sig[:960]);  seq.append(L.opus_multistream_encode(e1, p.ctypes.data, 960, buf, 4000)); seq.append(bytes(buf[:max(seq[-1], 0)]))
        p = np.ascontiguousarray(sig[960:1200]); seq.append(L.opus_multistream_encode(e2, p.ctypes.data, 240, buf, 4000)); seq.append(bytes(buf[:max(seq[-1], 0)]))
        p = np.ascontiguousarray(sig[1200:1440]); seq.append(L.opus_multistream_encode(e1, p.ctypes.data, 240, buf, 4000))          # RESTRICTED_SILK: OPUS_BAD_ARG
        p = np.ascontiguousarray(sig[1440:2400]); seq.append(L.opus_multistream_encode(e1, p.ctypes.data, 960, buf, 4000)); seq.append(bytes(buf[:max(seq[-1], 0)]))
        res.append(seq)
        L.opus_multistream_encoder_destroy(e1); L.opus_multistream_encoder_destroy(e2)
    assert res[0] == res[1]
    assert res[1][2] > 0 and res[1][4] == -1

def test_ms_hard_cbr_above_the_frame_cap():
    """opus_multistream_encode in hard CBR at 1.5 Mb/s over two mono streams: 3,750 bytes per 20 ms, the last stream padded past 1,275 bytes"""
    R, E = capi.load("ref"), capi.load(WHICH)
    for frame in (960, 1920):
        x = _ms_packet(R, 2, frame, bitrate=1500000, vbr=0, maxb=9000); y = _ms_packet(E, 2, frame, bitrate=1500000, vbr=0, maxb=9000)
        assert len(x) == 3750 * frame // 960 and x == y

def test_ms_decoder_gain_and_fec_frame_size_check():
    R, E = capi.load("ref"), capi.load(WHICH)
    pkts = [_ms_packet(R, 3, 960, seed=20 + i) for i in range(3)]
    out = []
    for L in (R, E):
        err = ci()
        d = _ms_proto(L).opus_multistream_decoder_create(48000, 3, 3, 0, bytes(range(3)), ctypes.byref(err))
        assert d and err.value == 0
        seq = []
        for i, g in enumerate((2560, -1280, 0)):
            assert _ms_set(L, d, 4034, g, enc=False) == 0
            pcm = np.zeros((960, 3), np.int16)
            n = L.opus_multistream_decode(d, pkts[i], len(pkts[i]), pcm.ctypes.data, 960, 0)
            seq.append((n, pcm.tobytes()))
        pcm = np.zeros((960, 3), np.int16)
        seq.append(L.opus_multistream_decode(d, pkts[0], len(pkts[0]), pcm.ctypes.data, 480, 1))     # decode_fec does not lift the frame_size check
        seq.append(L.opus_multistream_decode(d, pkts[0], len(pkts[0]), pcm.ctypes.data, 480, 0))
        out.append(seq)
        L.opus_multistream_decoder_destroy(d)
    assert out[0] == out[1]
    assert out[1][3] == -2 and out[1][4] == -2

def test_ms_decode_sub_packet_longer_than_six_frames():
    """a legal code-3 sub-packet of 48 CELT frames of 2.5 ms at ~300 bytes each (14 KB): decodes like the reference's"""
    R, E = capi.load("ref"), capi.load(WHICH)
    Fs = 48000
    a = capi.Enc("ref", Fs, 1, 2051, vbr=0, bitrate=960000)
    sig = sig_for(Fs, 1, 120 * 50, 61)
    R.opus_repacketizer_create.restype = vp; R.opus_repacketizer_cat.argtypes = [vp, ctypes.c_char_p, ci]; R.opus_repacketizer_out.argtypes = [vp, vp, ci]
    rp = R.opus_repacketizer_create()
    keep = []
    for i in range(48):
        p = a.encode(sig[i * 120:(i + 1) * 120], 120, 300)[0]
        keep.append(ctypes.create_string_buffer(p, len(p)))
        assert R.opus_repacketizer_cat(rp, keep[-1], len(p)) == 0
    big = (ctypes.c_ubyte * 20000)()
    n = R.opus_repacketizer_out(rp, big, 20000)
    assert n > 7696
    pkt = bytes(big[:n])
    out = []
    for L in (R, E):
        err = ci()
        d = _ms_proto(L).opus_multistream_decoder_create(Fs, 1, 1, 0, bytes([0]), ctypes.byref(err))
        pcm = np.zeros((5760, 1), np.int16)
        r = L.opus_multistream_decode(d, pkt, len(pkt), pcm.ctypes.data, 5760, 0)
        out.append((r, pcm.tobytes()))
        L.opus_multistream_decoder_destroy(d)
    assert out[0][0] == 5760 and out[0] == out[1]

def test_copy_and_import_of_stream_records_between_restricted_silk_and_other_applications_is_refused():
    """a RESTRICTED_SILK batch launches its back kernel without the CELT arena: records of VOIP / AUDIO batches (same kind, rate, channels) must not get there by
    opusgpu_enc_batch_copy_states or opusgpu_enc_batch_import_state, nor the other way round; like with like still works"""
    L = capi.load(WHICH)
    L.opusgpu_enc_batch_create.restype = vp; L.opusgpu_enc_batch_create.argtypes = [ctypes.c_int32, ctypes.c_int32, ci, ci, ci, ctypes.POINTER(ci)]
    L.opusgpu_enc_batch_copy_states.argtypes = [vp, ctypes.c_int32, vp, ctypes.c_int32, ctypes.c_int32]
    L.opusgpu_enc_batch_export_state.argtypes = [vp, ctypes.c_int32, vp]; L.opusgpu_enc_batch_import_state.argtypes = [vp, ctypes.c_int32, vp]
    L.opusgpu_enc_batch_destroy.argtypes = [vp]; L.opusgpu_enc_batch_destroy.restype = None
    err = ci()
    voip = L.opusgpu_enc_batch_create(2, 16000, 1, 2048, 0, ctypes.byref(err)); rs = L.opusgpu_enc_batch_create(2, 16000, 1, 2052, 0, ctypes.byref(err)); rs2 = L.opusgpu_enc_batch_create(2, 16000, 1, 2052, 0, ctypes.byref(err))
    assert voip and rs and rs2
    assert L.opusgpu_enc_batch_copy_states(rs, 0, voip, 0, 2) == -1 and L.opusgpu_enc_batch_copy_states(voip, 0, rs, 0, 2) == -1
    assert L.opusgpu_enc_batch_copy_states(rs2, 0, rs, 0, 2) == 0
    blob = (ctypes.c_ubyte * L.opusgpu_enc_sh_state_size())()
    assert L.opusgpu_enc_batch_export_state(voip, 0, blob) == 0
    assert L.opusgpu_enc_batch_import_state(rs, 1, blob) == -1 and L.opusgpu_enc_batch_import_state(voip, 1, blob) == 0
    assert L.opusgpu_enc_batch_export_state(rs, 0, blob) == 0
    assert L.opusgpu_enc_batch_import_state(voip, 1, blob) == -1 and L.opusgpu_enc_batch_import_state(rs2, 1, blob) == 0
    for b in (voip, rs, rs2): L.opusgpu_enc_batch_destroy(b)
