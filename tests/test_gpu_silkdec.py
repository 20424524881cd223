"""GPU: SILK-only packets from the compiled reference encoder through the C ABI (classic opus_decode and the batch decoder) against the compiled
reference decoder: identical PCM and OPUS_GET_FINAL_RANGE, state carried across packets."""
import numpy as np, pytest
from reflib import ref_fx
from test_oracle_encoder import RefEnc
from test_oracle_decoder import RefDec
from test_kernel_emu_silkdec import speechy

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(ref_fx() is None, reason="oracle/_ref not built")]

def _packets(enc_ch, frame, nframes, seed, **ctl):
    sig = speechy(nframes, enc_ch, seed, frame)
    e = RefEnc(enc_ch, application=2048, force_mode=1000, **ctl)
    out = []
    for i in range(nframes):
        pkt, n, erng = e.encode(np.ascontiguousarray(sig[i * frame:(i + 1) * frame]), frame)
        assert n > 1 and (pkt[0] >> 7) == 0
        out.append((pkt, erng))
    return out

@pytest.mark.parametrize("enc_ch,dec_ch,bw,bitrate,frame", [(1, 1, 1103, 24000, 960), (1, 1, 1102, 16000, 960), (1, 1, 1101, 10000, 960), (1, 1, 1103, 20000, 480),
                                                          (1, 1, 1103, 20000, 1920), (1, 1, 1103, 20000, 2880), (2, 2, 1103, 40000, 960), (2, 2, 1101, 20000, 960),
                                                          (1, 2, 1103, 20000, 960), (2, 1, 1103, 36000, 960)])
def test_gpu_silk_classic_decode(enc_ch, dec_ch, bw, bitrate, frame):
    import opus_amd
    pk = _packets(enc_ch, frame, 20, bw + bitrate + frame, bitrate=bitrate, bandwidth=bw)
    r = RefDec(dec_ch); d = opus_amd.OpusDecoder(48000, dec_ch)
    for i, (pkt, erng) in enumerate(pk):
        a = r.decode(pkt); pcm = d.decode(pkt, 5760)
        assert a[0] == frame == pcm.shape[0], (i, a[0], pcm.shape)
        assert d.final_range() == a[2] == erng, i
        assert np.array_equal(pcm.reshape(-1, dec_ch), a[1]), i

def test_gpu_silk_batch_decode_many_streams():
    """2,048 streams in one launch: 8 distinct SILK streams tiled; every replica must equal the reference decode of its source"""
    import opus_amd
    S = 2048; U = 8; nframes = 6
    pks = [_packets(1, 960, nframes, 100 + u, bitrate=14000 + 2000 * u, bandwidth=[1101, 1102, 1103][u % 3]) for u in range(U)]
    refs = [RefDec(1) for _ in range(U)]
    b = opus_amd.DecoderBatch(S, channels=1)
    for f in range(nframes):
        pcm, ns, rng = b.decode([pks[s % U][f][0] for s in range(S)], 960)
        for u in range(U):
            a = refs[u].decode(pks[u][f][0])
            sel = np.arange(u, S, U)
            assert (ns[sel] == 960).all() and (rng[sel] == a[2]).all(), (f, u)
            assert (pcm[sel, :, 0] == a[1][:, 0]).all(), (f, u)
    b.close()

def _mode_stream(enc_ch, frame, nframes, seed, schedule, **ctl):
    sig = speechy(nframes, enc_ch, seed, frame)
    e = RefEnc(enc_ch, application=2048, **ctl)
    req = dict(bitrate=4002, bandwidth=4008, max_bandwidth=4004, force_mode=11002, force_channels=4022)
    out = []
    for i in range(nframes):
        if i in schedule:
            for kk, v in schedule[i].items(): assert e.L.opus_encoder_ctl(e.st, req[kk], v) == 0
        pkt, n, erng = e.encode(np.ascontiguousarray(sig[i * frame:(i + 1) * frame]), frame)
        assert n > 1
        out.append((pkt, erng))
    return out

@pytest.mark.parametrize("ch,frame,bitrate,schedule", [
    (1, 960, 32000, {0: dict(force_mode=1001, bandwidth=1105)}),
    (2, 960, 48000, {0: dict(force_mode=1001, bandwidth=1104)}),
    (1, 480, 28000, {0: dict(force_mode=1001, bandwidth=1105)}),
    (1, 960, 28000, {0: dict(force_mode=1000, bandwidth=1103), 6: dict(force_mode=1001, bandwidth=1105), 12: dict(force_mode=1000, bandwidth=1103), 18: dict(force_mode=1001, bandwidth=1104)}),
    (2, 960, 44000, {0: dict(force_mode=1000, bandwidth=1103), 6: dict(force_mode=1001, bandwidth=1105), 12: dict(force_mode=1000, bandwidth=1102)}),
    (1, 960, 32000, {0: dict(force_mode=1002), 8: dict(force_mode=1000, bandwidth=1103)}),
    (2, 960, 48000, {0: dict(force_mode=1002), 8: dict(force_mode=1001, bandwidth=1105)})])
def test_gpu_hybrid_and_mode_switching(ch, frame, bitrate, schedule):
    """hybrid packets, SILK<->hybrid switches (CELT reset / silence-frame fade), CELT -> SILK/hybrid switches (redundancy frames, concealment cross-fade)"""
    import opus_amd
    pk = _mode_stream(ch, frame, 24, ch * 1000 + bitrate, schedule, bitrate=bitrate)
    r = RefDec(ch); d = opus_amd.OpusDecoder(48000, ch)
    for i, (pkt, erng) in enumerate(pk):
        a = r.decode(pkt); pcm = d.decode(pkt, 5760)
        assert a[0] == frame == pcm.shape[0], (i, a[0], pcm.shape)
        assert d.final_range() == a[2] == erng, i
        assert np.array_equal(pcm.reshape(-1, ch), a[1]), i

@pytest.mark.parametrize("ch,mode,bw,bitrate,frame", [(1, 1000, 1103, 20000, 960), (2, 1000, 1103, 36000, 960), (1, 1001, 1105, 32000, 960), (2, 1001, 1104, 44000, 960),
                                                    (1, 1000, 1102, 16000, 480), (1, 1000, 1101, 12000, 1920)])
def test_gpu_silk_packet_loss(ch, mode, bw, bitrate, frame):
    """lost packets in SILK-only and hybrid streams: SILK concealment + comfort noise (+ CELT noise concealment above 8 kHz), recovery glue"""
    import opus_amd
    pk = _mode_stream(ch, frame, 30, mode + bw + frame, {0: dict(force_mode=mode, bandwidth=bw)}, bitrate=bitrate)
    lose = {3, 7, 8, 9, 14, 15, 16, 17, 18, 19, 20, 24}
    r = RefDec(ch); d = opus_amd.OpusDecoder(48000, ch)
    for i, (pkt, erng) in enumerate(pk):
        if i in lose: pkt = b""
        a = r.decode(pkt, frame); pcm = d.decode(pkt if pkt else None, frame)
        assert a[0] == frame == pcm.shape[0], (i, a[0], pcm.shape)
        assert d.final_range() == a[2], i
        assert np.array_equal(pcm.reshape(-1, ch), a[1]), (i, i in lose)

def test_gpu_all_mode_transitions_with_loss():
    import opus_amd
    sched = {0: dict(force_mode=1000, bandwidth=1103), 8: dict(force_mode=1002, bandwidth=1105), 16: dict(force_mode=1001, bandwidth=1105), 24: dict(force_mode=1002)}
    for ch, lose in ((1, set()), (2, set()), (1, {7, 8, 15, 16, 23, 24, 25})):
        pk = _mode_stream(ch, 960, 32, 50 + ch, sched, bitrate=36000 * ch)
        r = RefDec(ch); d = opus_amd.OpusDecoder(48000, ch)
        for i, (pkt, erng) in enumerate(pk):
            if i in lose: pkt = b""
            a = r.decode(pkt, 960); pcm = d.decode(pkt if pkt else None, 960)
            assert a[0] == 960 == pcm.shape[0] and d.final_range() == a[2], (ch, i)
            assert np.array_equal(pcm.reshape(-1, ch), a[1]), (ch, i, i in lose)

def test_gpu_decode24_matches_reference():
    """opus_decode24 (the entry point opus_demo uses) on a hybrid stream, next to the reference's opus_decode24"""
    import ctypes, opus_amd
    L = opus_amd.lib(); R = ref_fx()
    pk = _mode_stream(2, 960, 10, 77, {0: dict(force_mode=1001, bandwidth=1105)}, bitrate=48000)
    err = ctypes.c_int()
    R.opus_decoder_create.restype = ctypes.c_void_p; R.opus_decoder_create.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    rd = R.opus_decoder_create(48000, 2, ctypes.byref(err)); d = opus_amd.OpusDecoder(48000, 2)
    R.opus_decode24.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    L.opus_decode24.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    for i, (pkt, _) in enumerate(pk):
        a = np.zeros((5760, 2), np.int32); b = np.zeros((5760, 2), np.int32)
        na = R.opus_decode24(rd, pkt, len(pkt), a.ctypes.data, 5760, 0); nb = L.opus_decode24(d._st, pkt, len(pkt), b.ctypes.data, 5760, 0)
        assert na == nb == 960 and np.array_equal(a[:na], b[:nb]), i

@pytest.mark.parametrize("ch,mode,bw,bitrate,frame", [(1, 1000, 1103, 24000, 960), (2, 1000, 1103, 40000, 960), (1, 1001, 1105, 36000, 960), (1, 1000, 1101, 14000, 1920)])
def test_gpu_inband_fec(ch, mode, bw, bitrate, frame):
    """lost packets recovered from the LBRR copy in the next packet (opus_decode(next, decode_fec = 1)), as a jitter buffer drives the reference"""
    import opus_amd
    sig = speechy(26, ch, mode + bw + frame + ch, frame)
    e = RefEnc(ch, application=2048, inband_fec=1, packet_loss=25, force_mode=mode, bandwidth=bw, bitrate=bitrate)
    pk = [e.encode(np.ascontiguousarray(sig[i * frame:(i + 1) * frame]), frame)[0] for i in range(26)]
    lose = {3, 7, 8, 12, 16, 17, 18, 22}
    r = RefDec(ch); d = opus_amd.OpusDecoder(48000, ch)
    for i in range(26):
        if i in lose:
            if i + 1 < 26 and (i + 1) not in lose: a = r.decode(pk[i + 1], frame, fec=1); pcm = d.decode(pk[i + 1], frame, 1)
            else: a = r.decode(b"", frame); pcm = d.decode(None, frame)
        else: a = r.decode(pk[i], frame); pcm = d.decode(pk[i], frame)
        assert a[0] == frame == pcm.shape[0] and d.final_range() == a[2], i
        assert np.array_equal(pcm.reshape(-1, ch), a[1]), (i, i in lose)
