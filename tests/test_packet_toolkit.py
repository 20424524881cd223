"""CPU: the host-side packet toolkit exported by libopus_amd.so (opus_packet_*, opus_repacketizer_*, (un)padding, multistream
(un)padding — pure host code, no GPU involved) against the same entry points of the compiled reference, on real packets, on
repacketised / padded packets and on mutated (fuzzed) bytes: identical return codes and identical output bytes."""
import ctypes, numpy as np, pytest
from reflib import ref_fx
import opus_amd, signals
from test_oracle_encoder import RefEnc

pytestmark = pytest.mark.skipif(ref_fx() is None, reason="compiled reference (oracle/_ref) not built")

def _libs():
    opus_amd.build()
    A = ctypes.CDLL(opus_amd.LIB_PATH); R = ref_fx()
    for L in (A, R):
        L.opus_repacketizer_create.restype = ctypes.c_void_p
        L.opus_repacketizer_init.restype = ctypes.c_void_p; L.opus_repacketizer_init.argtypes = [ctypes.c_void_p]
        L.opus_repacketizer_destroy.argtypes = [ctypes.c_void_p]
        L.opus_repacketizer_cat.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int]
        L.opus_repacketizer_out.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        L.opus_repacketizer_out_range.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
        L.opus_repacketizer_get_nb_frames.argtypes = [ctypes.c_void_p]
        L.opus_packet_parse.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        for f in ("opus_packet_get_bandwidth", "opus_packet_get_nb_channels"): getattr(L, f).argtypes = [ctypes.c_char_p]
        L.opus_packet_get_samples_per_frame.argtypes = [ctypes.c_char_p, ctypes.c_int]
        L.opus_packet_get_nb_frames.argtypes = [ctypes.c_char_p, ctypes.c_int]
        L.opus_packet_get_nb_samples.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int]
        L.opus_packet_pad.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        L.opus_packet_unpad.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.opus_multistream_packet_pad.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.opus_multistream_packet_unpad.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    return A, R

def _packets():
    out = []
    for ch, br, fr in [(2, 96000, 960), (1, 24000, 480), (2, 64000, 240), (2, 32000, 120), (2, 510000, 960)]:
        e = RefEnc(ch, bitrate=br, complexity=5)
        sig = signals.music(4, channels=ch, seed=br)
        out.append([e.encode(np.ascontiguousarray(sig[i * fr:(i + 1) * fr]), fr)[0] for i in range(4 * 960 // fr)])
    return out

def _parse(L, pkt):
    toc = ctypes.c_ubyte(); size = (ctypes.c_int16 * 48)(); off = ctypes.c_int(-1); frames = (ctypes.c_void_p * 48)()
    n = L.opus_packet_parse(pkt, len(pkt), ctypes.byref(toc), frames, size, ctypes.byref(off))
    base = ctypes.cast(ctypes.c_char_p(pkt), ctypes.c_void_p).value
    return (n, toc.value if n > 0 else None, list(size[:max(n, 0)]), off.value if n > 0 else None)

def _info(L, pkt):
    return (L.opus_packet_get_bandwidth(pkt), L.opus_packet_get_nb_channels(pkt), L.opus_packet_get_samples_per_frame(pkt, 48000), L.opus_packet_get_samples_per_frame(pkt, 16000),
            L.opus_packet_get_nb_frames(pkt, len(pkt)), L.opus_packet_get_nb_samples(pkt, len(pkt), 48000), L.opus_packet_get_nb_samples(pkt, len(pkt), 8000))

def _repack(L, pkts, ranges, maxlen=8000):
    rp = L.opus_repacketizer_create()
    res = []
    for p in pkts: res.append(L.opus_repacketizer_cat(rp, p, len(p)))
    res.append(L.opus_repacketizer_get_nb_frames(rp))
    for (b, e) in ranges:
        buf = (ctypes.c_ubyte * maxlen)()
        n = L.opus_repacketizer_out_range(rp, b, e, buf, maxlen)
        res.append((n, bytes(buf[:max(n, 0)])))
    buf = (ctypes.c_ubyte * maxlen)()
    n = L.opus_repacketizer_out(rp, buf, maxlen)
    res.append((n, bytes(buf[:max(n, 0)])))
    L.opus_repacketizer_destroy(rp)
    return res

def test_info_and_parse_on_real_and_fuzzed_packets():
    A, R = _libs()
    rng = np.random.default_rng(5)
    allp = [p for grp in _packets() for p in grp]
    for p in allp:
        assert _info(A, p) == _info(R, p) and _parse(A, p) == _parse(R, p)
    for k in range(3000):                      # arbitrary TOC / framing bytes in front of real payload or noise
        base = allp[k % len(allp)]
        n = int(rng.integers(1, 40))
        q = bytes(rng.integers(0, 256, n, dtype=np.uint8)) + (base if k % 2 else b"")
        q = q[:int(rng.integers(1, len(q) + 1))]
        assert _info(A, q) == _info(R, q), q[:8].hex()
        assert _parse(A, q) == _parse(R, q), q[:8].hex()

def test_repacketizer_and_padding():
    A, R = _libs()
    for grp in _packets():
        for g in (1, 2, 3, 4):
            if len(grp) < g: continue
            sub = grp[:g]
            ranges = [(0, 1), (0, g), (g - 1, g), (1, g), (0, g + 1), (1, 1)]
            a = _repack(A, sub, ranges); r = _repack(R, sub, ranges)
            assert a == r, (g, [x if isinstance(x, int) else x[0] for x in a], [x if isinstance(x, int) else x[0] for x in r])
            merged = r[-1][1]
            if not merged: continue
            a2 = _repack(A, [merged, merged], [(0, 2 * g)]); r2 = _repack(R, [merged, merged], [(0, 2 * g)])     # cat of multi-frame packets
            assert a2 == r2
            for tight in (len(merged) - 1, len(merged), len(merged) + 1):
                assert _repack(A, sub, [(0, g)], maxlen=max(tight, 1)) == _repack(R, sub, [(0, g)], maxlen=max(tight, 1))
            for new_len in (len(merged), len(merged) + 1, len(merged) + 2, len(merged) + 255, len(merged) + 256, len(merged) + 700, len(merged) - 1):
                outs = []
                for L in (A, R):
                    buf = (ctypes.c_ubyte * 12000)(*merged)
                    rc = L.opus_packet_pad(buf, len(merged), new_len)
                    padded = bytes(buf[:new_len]) if rc == 0 else b""
                    un = -99; unb = b""
                    if rc == 0:
                        buf2 = (ctypes.c_ubyte * 12000)(*padded)
                        un = L.opus_packet_unpad(buf2, new_len); unb = bytes(buf2[:max(un, 0)])
                    outs.append((rc, padded, un, unb))
                assert outs[0] == outs[1], (g, new_len, outs[0][0], outs[1][0])

def test_multistream_pad_unpad():
    A, R = _libs()
    grp = _packets()
    rp = R.opus_repacketizer_create()
    # build a 3-stream multistream packet with the reference: streams 0,1 self-delimited (via out_range on a padded copy), last one plain
    def selfdelim(p):           # code-0 packet -> self-delimited framing: toc, size, payload
        L0 = len(p) - 1
        sz = bytes([L0]) if L0 < 252 else bytes([252 + (L0 & 3), (L0 - (252 + (L0 & 3))) >> 2])
        return p[:1] + sz + p[1:]
    ms = selfdelim(grp[0][0]) + selfdelim(grp[1][0]) + grp[2][0]
    for new_len in (len(ms), len(ms) + 1, len(ms) + 3, len(ms) + 300):
        outs = []
        for L in (A, R):
            buf = (ctypes.c_ubyte * 12000)(*ms)
            rc = L.opus_multistream_packet_pad(buf, len(ms), new_len, 3)
            padded = bytes(buf[:new_len]) if rc == 0 else b""
            buf2 = (ctypes.c_ubyte * 12000)(*(padded if rc == 0 else ms))
            un = L.opus_multistream_packet_unpad(buf2, len(padded) if rc == 0 else len(ms), 3)
            outs.append((rc, padded, un, bytes(buf2[:max(un, 0)])))
        assert outs[0] == outs[1], (new_len, outs[0][0], outs[1][0], outs[0][2], outs[1][2])
    R.opus_repacketizer_destroy(rp)
