"""GPU (MI355X): the HIP decoder, called through the C ABI, must reproduce the reference bit for bit: PCM, sample counts and
OPUS_GET_FINAL_RANGE against the oracle decoder and (where it travelled) the compiled reference decoder; encoder final range ==
decoder final range; full-size (65,536 streams) encode -> decode round trip entirely on the device buffers' host mirrors."""
import ctypes, numpy as np, pytest
import signals
from reflib import oracle, ref_fx

pytestmark = pytest.mark.gpu

def _oa():
    import opus_amd
    return opus_amd

def _check(S, frames, enc_ch, dec_ch, frame, ctl, checker="oracle"):
    oa = _oa()
    from test_oracle_encoder import OracleEnc
    from test_oracle_decoder import OracleDec, RefDec
    encs = [OracleEnc(enc_ch, **ctl) for _ in range(S)]
    Dec = OracleDec if checker == "oracle" else RefDec
    chk = [Dec(dec_ch) for _ in range(S)]
    b = oa.DecoderBatch(S, channels=dec_ch)
    sigs = [signals.music(frames * frame // 960 + 1, channels=enc_ch, seed=200 + s) if s % 3 else signals.noise_bursts(frames * frame // 960 + 1, channels=enc_ch, seed=s) for s in range(S)]
    for i in range(frames):
        pk = []; er = []
        for s in range(S):
            p, n, r = encs[s].encode(np.ascontiguousarray(sigs[s][i * frame:(i + 1) * frame]), frame)
            pk.append(p); er.append(r)
        pcm, ns, rng = b.decode(pk, frame)
        for s in range(S):
            a = chk[s].decode(pk[s])
            assert a[0] == int(ns[s]) == frame, (i, s, a[0], int(ns[s]))
            assert a[2] == int(rng[s]) == er[s], (i, s, hex(a[2]), hex(int(rng[s])))
            assert np.array_equal(a[1], pcm[s, :frame]), (i, s, np.nonzero(a[1] != pcm[s, :frame])[0][:6])
    b.close()

def test_gpu_dec_config2_vs_oracle():
    _check(32, 40, 2, 2, 960, dict(bitrate=128000, complexity=10))

@pytest.mark.skipif(ref_fx() is None, reason="compiled reference did not travel")
def test_gpu_dec_config2_vs_reference():
    _check(12, 30, 2, 2, 960, dict(bitrate=128000, complexity=10), checker="ref")

@pytest.mark.parametrize("enc_ch,dec_ch,bitrate,complexity,frame", [
    (2, 2, 64000, 10, 960), (2, 2, 24000, 10, 960), (2, 2, 510000, 10, 960), (1, 1, 64000, 10, 960), (1, 1, 12000, 5, 960),
    (2, 2, 128000, 10, 480), (2, 2, 128000, 10, 240), (2, 2, 128000, 10, 120), (2, 2, 8000, 10, 960), (1, 2, 48000, 10, 960), (2, 1, 96000, 10, 960)])
def test_gpu_dec_rates_sizes(enc_ch, dec_ch, bitrate, complexity, frame):
    _check(6, min(16 * 960 // frame, 40), enc_ch, dec_ch, frame, dict(bitrate=bitrate, complexity=complexity))

def test_gpu_dec_bandwidths_and_edge_inputs():
    for bw in (1101, 1103, 1104):
        _check(3, 8, 2, 2, 960, dict(bitrate=64000, complexity=10, bandwidth=bw))
    oa = _oa()
    from test_oracle_encoder import OracleEnc
    from test_oracle_decoder import OracleDec
    kinds = [signals.tone(12, 440.0), signals.silence_then_music(12, seed=3), (signals.music(12, seed=4).astype(np.int32) * 4).clip(-32768, 32767).astype(np.int16),
             np.zeros((12 * 960, 2), np.int16), np.full((12 * 960, 2), 32767, np.int16)]
    S = len(kinds)
    encs = [OracleEnc(2, bitrate=96000, complexity=10) for _ in range(S)]; chk = [OracleDec(2) for _ in range(S)]
    b = oa.DecoderBatch(S, channels=2)
    for i in range(12):
        pk = [encs[s].encode(np.ascontiguousarray(kinds[s][i * 960:(i + 1) * 960]), 960)[0] for s in range(S)]
        pcm, ns, rng = b.decode(pk, 960)
        for s in range(S):
            a = chk[s].decode(pk[s])
            assert a[0] == int(ns[s]) and a[2] == int(rng[s]) and np.array_equal(a[1], pcm[s, :a[0]]), (i, s)
    b.close()

@pytest.mark.skipif(ref_fx() is None, reason="compiled reference did not travel")
def test_gpu_dec_multiframe_packets_and_errors():
    """code-1/2/3 packets from the reference repacketizer; per-stream error codes for bad / unsupported packets"""
    oa = _oa()
    from test_oracle_encoder import RefEnc
    from test_oracle_decoder import RefDec
    L = ref_fx()
    L.opus_repacketizer_create.restype = ctypes.c_void_p
    L.opus_repacketizer_cat.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int]
    L.opus_repacketizer_out.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    L.opus_repacketizer_init.argtypes = [ctypes.c_void_p]; L.opus_repacketizer_init.restype = ctypes.c_void_p
    rp = L.opus_repacketizer_create()
    for frame, group, vbr in [(960, 2, 1), (480, 3, 1), (240, 6, 1), (480, 2, 0)]:
        sig = signals.music(24, seed=11)
        e = RefEnc(2, bitrate=96000, complexity=10, vbr=vbr); r = RefDec(2)
        b = oa.DecoderBatch(2, channels=2)
        for g in range(6):
            L.opus_repacketizer_init(rp)
            for k in range(group):
                i = g * group + k
                pkt, n, _ = e.encode(np.ascontiguousarray(sig[i * frame:(i + 1) * frame]), frame)
                assert L.opus_repacketizer_cat(rp, pkt, len(pkt)) == 0
            out = (ctypes.c_ubyte * 8000)()
            m = L.opus_repacketizer_out(rp, out, 8000)
            big = bytes(out[:m])
            a = r.decode(big)
            pcm, ns, rng = b.decode([big, big], frame * group)
            for s in range(2):
                assert a[0] == int(ns[s]) == frame * group and a[2] == int(rng[s]) and np.array_equal(a[1], pcm[s, :a[0]]), (frame, group, g, s)
        b.close()
    # error codes are per stream; a bad packet in one stream does not disturb its neighbours
    e = RefEnc(2, bitrate=64000); r = RefDec(2)
    good = e.encode(np.ascontiguousarray(signals.music(1, seed=1)[:960]), 960)[0]
    b = oa.DecoderBatch(4, channels=2)
    pcm, ns, rng = b.decode([good, b"\xfd\x01", b"\x08" + b"\0" * 20, good], 960)        # ok, invalid code-1 (odd length), a (garbage) SILK-only packet, ok
    a = r.decode(good)
    assert int(ns[0]) == int(ns[3]) == 960 and np.array_equal(pcm[0], a[1]) and np.array_equal(pcm[3], a[1])
    assert int(ns[1]) == -4
    a2 = RefDec(2).decode(b"\x08" + b"\0" * 20)                                          # a SILK-only packet decodes since the SILK decoder exists
    assert int(ns[2]) == a2[0] == 960 and np.array_equal(pcm[2], a2[1]) and int(rng[2]) == a2[2]
    pcm, ns, rng = b.decode([good] * 4, 480)
    assert all(int(x) == -2 for x in ns)                                                 # OPUS_BUFFER_TOO_SMALL
    b.close()

def test_gpu_dec_classic_api_and_state_contract():
    """classic opus_decoder_* entry points (batch of one) + the memcpy contract on the decoder blob; state export/import on the batch"""
    oa = _oa()
    from test_oracle_encoder import OracleEnc
    from test_oracle_decoder import OracleDec
    L = oa.lib()
    sig = signals.music(12, seed=21)
    e = OracleEnc(2, bitrate=96000, complexity=10); o = OracleDec(2)
    d = oa.OpusDecoder(48000, 2)
    pk = [e.encode(np.ascontiguousarray(sig[i * 960:(i + 1) * 960]), 960)[0] for i in range(12)]
    for i in range(4):
        a = o.decode(pk[i]); y = d.decode(pk[i], 960)
        assert np.array_equal(a[1], y) and a[2] == d.final_range()
    size = L.opus_decoder_get_size(2)
    blob = ctypes.create_string_buffer(size)
    ctypes.memmove(blob, d._st, size)
    ctypes.memset(d._st, 0xFF, size)                    # poison the original (tests/test_opus_decode.c:86-94)
    L.opus_decode.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    for i in range(4, 8):
        a = o.decode(pk[i])
        out = np.zeros((960, 2), np.int16)
        n = L.opus_decode(ctypes.cast(blob, ctypes.c_void_p), pk[i], len(pk[i]), out.ctypes.data, 960, 0)
        assert n == 960 and np.array_equal(a[1], out)
    d._st = None
    # batch export -> import into another slot -> continue
    b = oa.DecoderBatch(2, channels=2)
    o2 = OracleDec(2)
    for i in range(3):
        b.decode([pk[i], pk[i]], 960); o2.decode(pk[i])
    st = b.export_state(0)
    b.reset()
    b.import_state(1, st)
    for i in range(3, 6):
        pcm, ns, rng = b.decode([pk[0], pk[i]], 960)
        a = o2.decode(pk[i])
        assert int(ns[1]) == 960 and np.array_equal(a[1], pcm[1]) and a[2] == int(rng[1])
    b.close()
    assert L.opus_decode(ctypes.c_void_p(0), pk[0], len(pk[0]), None, 960, 0) == -1

def test_gpu_full_size_encode_decode_roundtrip():
    """BASELINE size: 65,536 streams encoded by the HIP encoder and decoded by the HIP decoder; encoder final range == decoder final
    range on EVERY stream and frame (the reference's own invariant, tests/test_opus_encode.c:499-501); a sampled subset is compared
    with the oracle decoder sample for sample."""
    oa = _oa()
    from test_oracle_decoder import OracleDec
    S = 65536
    eb = oa.EncoderBatch(S, channels=2); eb.ctl(oa.OPUS_SET_BITRATE_REQUEST, 128000); eb.ctl(oa.OPUS_SET_COMPLEXITY_REQUEST, 10)
    db = oa.DecoderBatch(S, channels=2)
    base = [signals.music(4, seed=k) for k in range(16)]
    sample = [0, 5, 15, 16 * 1000 + 3, 16 * 4095 + 15]
    od = {s: OracleDec(2) for s in sample}
    for i in range(3):
        fr = np.stack([base[k][i * 960:(i + 1) * 960].reshape(-1) for k in range(16)])
        pk, lens, erng = eb.encode(np.tile(fr, (S // 16, 1)), 960)
        pcm, ns, drng = db.decode(pk, 960)
        assert np.all(ns == 960)
        assert np.array_equal(erng, drng)
        for k in range(16): assert np.all(pcm[k::16] == pcm[k])           # identical streams decode identically on every wavefront
        for s in sample:
            a = od[s].decode(pk[s])
            assert a[0] == 960 and a[2] == int(drng[s]) and np.array_equal(a[1], pcm[s])
    eb.close(); db.close()

def test_gpu_dec_corrupted_packets():
    """bit flips, truncations, overwritten and random payloads: same PCM, sample count / error code and final range as the oracle on every
    stream, with the (now garbage-driven) state carried from packet to packet; neighbours in the batch are unaffected"""
    oa = _oa()
    from test_oracle_encoder import OracleEnc
    from test_oracle_decoder import OracleDec, _mutations
    rng = np.random.default_rng(31)
    S = 24
    e = OracleEnc(2, bitrate=96000, complexity=5)
    chk = [OracleDec(2) for _ in range(S)]
    b = oa.DecoderBatch(S, channels=2)
    sig = signals.music(30, seed=32)
    for i in range(30):
        pkt = e.encode(np.ascontiguousarray(sig[i * 960:(i + 1) * 960]), 960)[0]
        pk = _mutations(pkt, rng, S - 1) + [pkt]
        pcm, ns, rngs = b.decode(pk, 960)
        for s in range(S):
            a = chk[s].decode(pk[s])
            assert a[0] == int(ns[s]), (i, s, a[0], int(ns[s]), pk[s][:4].hex())
            if a[0] > 0: assert a[2] == int(rngs[s]) and np.array_equal(a[1], pcm[s, :a[0]]), (i, s)
    b.close()

@pytest.mark.parametrize("channels,bitrate,frame,pattern", [
    (2, 96000, 960, "single"), (2, 96000, 960, "burst"), (1, 32000, 960, "burst"), (2, 64000, 480, "random"), (2, 128000, 240, "random"),
    (2, 128000, 120, "burst"), (2, 48000, 960, "long"), (1, 64000, 480, "start")])
def test_gpu_dec_packet_loss(channels, bitrate, frame, pattern):
    """lost packets (lens[s] = 0) per stream: pitch PLC, fade, noise PLC after long bursts, recovery frames — identical to the oracle (== reference);
    every stream of the batch has its own loss pattern"""
    oa = _oa()
    from test_oracle_encoder import OracleEnc
    from test_oracle_decoder import OracleDec
    rng = np.random.default_rng(41)
    S = 8
    n = min(40 * 960 // frame, 100)
    sig = signals.music(n * frame // 960 + 1, channels=channels, seed=42)
    e = OracleEnc(channels, bitrate=bitrate, complexity=5)
    base = {"single": {10, 20, 30}, "burst": set(range(8, 14)) | set(range(30, 33)), "random": set(np.nonzero(rng.random(n) < 0.2)[0].tolist()),
            "long": set(range(6, 40)), "start": {0, 1, 5}}[pattern]
    lost = [{(i + 3 * s) % n for i in base} if s else base for s in range(S)]
    chk = [OracleDec(channels) for _ in range(S)]
    b = oa.DecoderBatch(S, channels=channels)
    for i in range(n):
        pkt = e.encode(np.ascontiguousarray(sig[i * frame:(i + 1) * frame]), frame)[0]
        pk = [b"" if i in lost[s] else pkt for s in range(S)]
        pcm, ns, rngs = b.decode(pk, frame)
        for s in range(S):
            if i in lost[s]:
                pb = np.zeros((frame, channels), np.int16)
                nb = chk[s].O.oc_opus_decode(chk[s].buf, None, 0, pb.ctypes.data, frame, 0)
                assert nb == int(ns[s]) == frame and int(rngs[s]) == 0 and np.array_equal(pb, pcm[s]), (i, s, "lost")
            else:
                a = chk[s].decode(pkt)
                assert a[0] == int(ns[s]) == frame and a[2] == int(rngs[s]) and np.array_equal(a[1], pcm[s]), (i, s, "recv")
    b.close()

@pytest.mark.skipif(ref_fx() is None, reason="compiled reference did not travel")
def test_gpu_dec_classic_loss_api_vs_reference():
    """opus_decode(st, NULL, 0, ...) and decode_fec = 1 through the classic entry point, against the compiled reference"""
    oa = _oa()
    from test_oracle_encoder import RefEnc
    from test_oracle_decoder import RefDec
    L = oa.lib()
    L.opus_decode.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    e = RefEnc(2, bitrate=64000, complexity=5); r = RefDec(2); d = oa.OpusDecoder(48000, 2)
    sig = signals.music(30, seed=51)
    for i in range(30):
        pkt = e.encode(np.ascontiguousarray(sig[i * 960:(i + 1) * 960]), 960)[0]
        pa = np.zeros((960, 2), np.int16); pb = np.zeros((960, 2), np.int16)
        if i % 7 == 3:
            na = r.L.opus_decode(r.st, None, 0, pa.ctypes.data, 960, 0); nb = L.opus_decode(d._st, None, 0, pb.ctypes.data, 960, 0)
        elif i % 7 == 5:
            na = r.L.opus_decode(r.st, pkt, len(pkt), pa.ctypes.data, 960, 1); nb = L.opus_decode(d._st, pkt, len(pkt), pb.ctypes.data, 960, 1)
        else:
            na = r.L.opus_decode(r.st, pkt, len(pkt), pa.ctypes.data, 960, 0); nb = L.opus_decode(d._st, pkt, len(pkt), pb.ctypes.data, 960, 0)
        assert na == nb == 960 and np.array_equal(pa, pb), (i, na, nb)
    assert L.opus_decode(d._st, None, 0, pb.ctypes.data, 961, 0) == r.L.opus_decode(r.st, None, 0, pa.ctypes.data, 961, 0) == -1


# ---- the same comparisons with the decoder's PVQ stage forced (oa_celt_dpvq_kernel: the bands of four streams per wave, opus_amd/csrc/celt_dec_pvq4.h; wide calls take it by
# default, these few-stream batches only through opusgpu_dec_batch_set_pvq_stage(b, 1)) ----
@pytest.fixture
def pvq_forced():
    oa = _oa(); old = oa.DEC_PVQ_STAGE_DEFAULT; oa.DEC_PVQ_STAGE_DEFAULT = 1
    yield
    oa.DEC_PVQ_STAGE_DEFAULT = old

def test_gpu_dec_pvq_stage_config2_vs_oracle(pvq_forced):
    _check(32, 40, 2, 2, 960, dict(bitrate=128000, complexity=10))

@pytest.mark.skipif(ref_fx() is None, reason="compiled reference did not travel")
def test_gpu_dec_pvq_stage_config2_vs_reference(pvq_forced):
    _check(12, 30, 2, 2, 960, dict(bitrate=128000, complexity=10), checker="ref")

@pytest.mark.parametrize("enc_ch,dec_ch,bitrate,complexity,frame", [
    (2, 2, 64000, 10, 960), (2, 2, 24000, 10, 960), (2, 2, 510000, 10, 960), (1, 1, 64000, 10, 960), (1, 1, 12000, 5, 960), (1, 1, 256000, 10, 480),
    (2, 2, 128000, 10, 480), (2, 2, 8000, 10, 960), (1, 2, 48000, 10, 960), (2, 1, 96000, 10, 960)])
def test_gpu_dec_pvq_stage_rates_sizes(pvq_forced, enc_ch, dec_ch, bitrate, complexity, frame):
    _check(6, min(16 * 960 // frame, 40), enc_ch, dec_ch, frame, dict(bitrate=bitrate, complexity=complexity))

@pytest.mark.parametrize("channels,bitrate,frame,pattern", [(2, 96000, 960, "burst"), (2, 64000, 480, "random"), (1, 64000, 480, "start")])
def test_gpu_dec_pvq_stage_packet_loss(pvq_forced, channels, bitrate, frame, pattern):
    """streams leave the pipeline for the general kernel on a loss (and for the frame after it) and come back"""
    test_gpu_dec_packet_loss(channels, bitrate, frame, pattern)
