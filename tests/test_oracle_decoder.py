"""CPU: whole-decoder parity of the plain-C restatement (oracle/oc_celt_dec.c, oc_opus_dec.c) against the compiled reference
(fixed-point build): identical PCM and OPUS_GET_FINAL_RANGE, packet after packet with state carried; decoder final range ==
encoder final range (the invariant the reference's own tests check, tests/test_opus_encode.c:499-501)."""
import ctypes, numpy as np, pytest
from reflib import ref_fx, oracle
import signals
from test_oracle_encoder import RefEnc

pytestmark = pytest.mark.skipif(ref_fx() is None or oracle() is None, reason="oracle/_ref or oracle lib not built")

class RefDec:
    def __init__(self, channels):
        L = self.L = ref_fx()
        L.opus_decoder_create.restype = ctypes.c_void_p
        L.opus_decoder_create.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
        L.opus_decode.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        L.opus_decoder_ctl.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        err = ctypes.c_int()
        self.st = L.opus_decoder_create(48000, channels, ctypes.byref(err))
        assert err.value == 0
        self.ch = channels
    def decode(self, pkt, max_frame=5760, fec=0):
        pcm = np.zeros((max_frame, self.ch), np.int16)
        n = self.L.opus_decode(self.st, pkt, len(pkt), pcm.ctypes.data, max_frame, int(fec))
        rng = ctypes.c_uint32()
        self.L.opus_decoder_ctl(self.st, 4031, ctypes.byref(rng))
        return n, pcm[:max(n, 0)].copy(), rng.value

class OracleDec:
    def __init__(self, channels):
        O = self.O = oracle()
        self.buf = ctypes.create_string_buffer(O.oc_opus_dec_size())
        assert O.oc_opus_dec_init(self.buf, 48000, channels) == 0
        O.oc_opus_decode.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        O.oc_opus_dec_final_range.restype = ctypes.c_uint32
        self.ch = channels
    def decode(self, pkt, max_frame=5760):
        pcm = np.zeros((max_frame, self.ch), np.int16)
        n = self.O.oc_opus_decode(self.buf, pkt, len(pkt), pcm.ctypes.data, max_frame, 0)
        return n, pcm[:max(n, 0)].copy(), self.O.oc_opus_dec_final_range(self.buf)

def _roundtrip(enc_ch, dec_ch, sig, frame, nframes, **ctl):
    e = RefEnc(enc_ch, **ctl); r = RefDec(dec_ch); o = OracleDec(dec_ch)
    for i in range(nframes):
        pcm = np.ascontiguousarray(sig[i * frame:(i + 1) * frame])
        pkt, n, erng = e.encode(pcm, frame)
        assert n > 0
        a = r.decode(pkt); b = o.decode(pkt)
        assert a[0] == b[0] == frame, (i, a[0], b[0])
        assert a[2] == b[2] == erng, (i, hex(a[2]), hex(b[2]), hex(erng))
        assert np.array_equal(a[1], b[1]), (i, np.nonzero(a[1] != b[1])[0][:5])

@pytest.mark.parametrize("seed", range(3))
def test_dec_config2_stereo_128k(seed):
    _roundtrip(2, 2, signals.music(30, seed=seed), 960, 30, bitrate=128000, complexity=10)

@pytest.mark.parametrize("channels,bitrate,complexity,frame", [
    (2, 64000, 10, 960), (2, 24000, 10, 960), (2, 510000, 10, 960), (1, 64000, 10, 960), (1, 12000, 5, 960), (2, 96000, 5, 960),
    (2, 48000, 3, 960), (2, 128000, 10, 480), (2, 128000, 10, 240), (2, 128000, 10, 120), (1, 48000, 10, 480), (2, 16000, 10, 960), (2, 8000, 10, 960)])
def test_dec_rates_sizes(channels, bitrate, complexity, frame):
    n = min(24 * 960 // frame, 60)
    _roundtrip(channels, channels, signals.music(24, channels=channels, seed=7), frame, n, bitrate=bitrate, complexity=complexity)

@pytest.mark.parametrize("kind", ["bursts", "tone", "silence", "loud"])
def test_dec_signal_kinds(kind):
    sig = {"bursts": signals.noise_bursts(30, seed=2), "tone": signals.tone(30, 440.0), "silence": signals.silence_then_music(30, seed=3),
           "loud": (signals.music(30, seed=4).astype(np.int32) * 4).clip(-32768, 32767).astype(np.int16)}[kind]
    _roundtrip(2, 2, sig, 960, 30, bitrate=96000, complexity=10)

def test_dec_channel_mismatch_and_bandwidths():
    _roundtrip(1, 2, signals.music(20, channels=1, seed=5), 960, 20, bitrate=48000, complexity=10)      # mono stream -> stereo decoder
    _roundtrip(2, 1, signals.music(20, seed=6), 960, 20, bitrate=96000, complexity=10)                   # stereo stream -> mono decoder
    for bw in (1101, 1103, 1104):
        _roundtrip(2, 2, signals.music(12, seed=8), 960, 12, bitrate=64000, complexity=10, bandwidth=bw)
    _roundtrip(2, 2, signals.music(12, seed=9), 960, 12, bitrate=32000, complexity=10, force_channels=1)

def test_dec_multiframe_packets():
    """code-1/2/3 packets built with the reference repacketizer (src/repacketizer.c) from 2..6 CELT frames"""
    L = ref_fx()
    L.opus_repacketizer_create.restype = ctypes.c_void_p
    L.opus_repacketizer_cat.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int]
    L.opus_repacketizer_out.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    L.opus_repacketizer_init.argtypes = [ctypes.c_void_p]; L.opus_repacketizer_init.restype = ctypes.c_void_p
    rp = L.opus_repacketizer_create()
    for frame, group, vbr in [(960, 2, 1), (480, 3, 1), (240, 6, 1), (480, 2, 0)]:
        sig = signals.music(24, seed=11)
        e = RefEnc(2, bitrate=96000, complexity=10, vbr=vbr); r = RefDec(2); o = OracleDec(2)
        nfr = 24 * 960 // frame // group
        for g in range(min(nfr, 10)):
            L.opus_repacketizer_init(rp)
            keep = []
            for k in range(group):
                i = g * group + k
                pkt, n, _ = e.encode(np.ascontiguousarray(sig[i * frame:(i + 1) * frame]), frame)
                keep.append(pkt)
                assert L.opus_repacketizer_cat(rp, pkt, len(pkt)) == 0
            out = (ctypes.c_ubyte * 8000)()
            m = L.opus_repacketizer_out(rp, out, 8000)
            assert m > 0
            big = bytes(out[:m])
            a = r.decode(big); b = o.decode(big)
            assert a[0] == b[0] == frame * group and a[2] == b[2] and np.array_equal(a[1], b[1]), (frame, group, g, a[0], b[0])

def test_dec_error_codes():
    o = OracleDec(2); r = RefDec(2)
    for pkt in (b"", b"\xfc", b"\xff\x40", b"\xfd\x01"):
        if len(pkt) == 0: continue
        a = r.decode(pkt, 960); b = o.decode(pkt, 960)
        if a[0] < 0: assert b[0] == a[0] or b[0] == -5, (pkt, a[0], b[0])
    e = RefEnc(2, bitrate=64000)
    pkt, n, _ = e.encode(np.ascontiguousarray(signals.music(1, seed=1)[:960]), 960)
    assert r.decode(pkt, 480)[0] == o.decode(pkt, 480)[0] == -2          # OPUS_BUFFER_TOO_SMALL

def _mutations(pk, rng, n):
    """corrupted variants of a real packet: bit flips, truncation, byte overwrite, random tail; the TOC stays CELT-only"""
    out = []
    for _ in range(n):
        b = bytearray(pk)
        kind = rng.integers(0, 4)
        if kind == 0:
            for _ in range(int(rng.integers(1, 6))):
                i = int(rng.integers(1, len(b))); b[i] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:
            b = b[:max(3, int(rng.integers(2, len(b))))]
        elif kind == 2:
            i = int(rng.integers(1, len(b))); j = min(len(b), i + int(rng.integers(1, 12)))
            b[i:j] = bytes(rng.integers(0, 256, j - i, dtype=np.uint8))
        else:
            b = bytearray(b[:1]) + bytes(rng.integers(0, 256, int(rng.integers(2, 300)), dtype=np.uint8))
        b[0] = (b[0] & 0xfc) | 0x80                      # keep it a single-frame CELT-only packet
        out.append(bytes(b))
    return out

def test_dec_corrupted_packets_match_reference():
    """the decoder is deterministic on ANY bytes: corrupted packets must decode to the same PCM / final range as the reference does"""
    rng = np.random.default_rng(11)
    e = RefEnc(2, bitrate=96000, complexity=5); r = RefDec(2); o = OracleDec(2)
    sig = signals.music(40, seed=12)
    for i in range(40):
        pkt, n, _ = e.encode(np.ascontiguousarray(sig[i * 960:(i + 1) * 960]), 960)
        for q in _mutations(pkt, rng, 6) + [pkt]:
            a = r.decode(q); b = o.decode(q)
            assert a[0] == b[0], (i, a[0], b[0], q[:4].hex())
            if a[0] > 0: assert a[2] == b[2] and np.array_equal(a[1], b[1]), (i, q[:4].hex())

def _decode_loss(dec, L_decode, frame):
    pcm = np.zeros((frame, dec.ch), np.int16)
    n = L_decode(None, 0, pcm.ctypes.data, frame, 0)
    return n, pcm[:max(n, 0)].copy()

@pytest.mark.parametrize("channels,bitrate,frame,pattern", [
    (2, 96000, 960, "single"), (2, 96000, 960, "burst"), (1, 32000, 960, "burst"), (2, 64000, 480, "random"), (2, 128000, 240, "random"),
    (2, 128000, 120, "burst"), (2, 48000, 960, "long"), (1, 64000, 480, "start")])
def test_dec_packet_loss_concealment(channels, bitrate, frame, pattern):
    """opus_decode(NULL): pitch-based PLC, its fade, the switch to noise PLC after 40+ lost 2.5 ms units, recovery frames (prefilter_and_fold,
    energy safety) — PCM identical to the reference; state carried across losses."""
    rng = np.random.default_rng(41)
    n = min(60 * 960 // frame, 160)
    sig = signals.music(n * frame // 960 + 1, channels=channels, seed=42)
    e = RefEnc(channels, bitrate=bitrate, complexity=5); r = RefDec(channels); o = OracleDec(channels)
    lost = {"single": {10, 20, 30}, "burst": set(range(8, 14)) | set(range(30, 33)), "random": set(np.nonzero(rng.random(n) < 0.2)[0].tolist()),
            "long": set(range(6, 40)), "start": {0, 1, 5}}[pattern]
    for i in range(n):
        pkt, m, erng = e.encode(np.ascontiguousarray(sig[i * frame:(i + 1) * frame]), frame)
        if i in lost:
            pa = np.zeros((frame, channels), np.int16); pb = np.zeros((frame, channels), np.int16)
            na = r.L.opus_decode(r.st, None, 0, pa.ctypes.data, frame, 0)
            nb = o.O.oc_opus_decode(o.buf, None, 0, pb.ctypes.data, frame, 0)
            assert na == nb == frame, (i, na, nb)
            assert np.array_equal(pa, pb), (i, "lost", np.argwhere(pa != pb)[:4])
        else:
            a = r.decode(pkt); b = o.decode(pkt)
            assert a[0] == b[0] == frame and a[2] == b[2] and np.array_equal(a[1], b[1]), (i, "recv", np.argwhere(a[1] != b[1])[:4])
