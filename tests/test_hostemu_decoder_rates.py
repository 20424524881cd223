"""CPU: the classic decoder API of the product's C ABI on the wave emulator at every API rate (8-48 kHz) against the compiled reference decoder:
CELT down-sampling in the de-emphasis (celt/celt_decoder.c:318), SILK resampling to the API rate, hybrid, losses, mode switches."""
import numpy as np, pytest
import capi, signals
from reflib import ref_fx
from test_kernel_emu_silkdec import speechy
pytestmark = pytest.mark.skipif(ref_fx() is None, reason="oracle/_ref not built")
WHICH = "emu"

def _stream(app, ch, frame, nframes, schedule=None, seed=0, **ctl):
    """packets of the reference encoder at 48 kHz"""
    sig = speechy(nframes, ch, seed, frame) if app != 2051 else signals.music(nframes, frame, ch, seed)
    e = capi.Enc("ref", 48000, ch, app, **ctl)
    pk = []
    for i in range(nframes):
        if schedule and i in schedule:
            for k, v in schedule[i].items(): assert e.set(k, v) == 0
        p, n, rng = e.encode(sig[i * frame:(i + 1) * frame], frame)
        assert n > 0
        pk.append(p)
    return pk

def _compare(pk, Fs, dec_ch, lose=()):
    a = capi.Dec("ref", Fs, dec_ch); b = capi.Dec(WHICH, Fs, dec_ch)
    for i, p in enumerate(pk):
        nsamp = (len(p) and 1) and None
        if i in lose:
            n = Fs // 50
            x = a.decode(b"", n); y = b.decode(b"", n)
        else:
            x = a.decode(p); y = b.decode(p)
        assert x[0] == y[0], (Fs, i, x[0], y[0])
        assert x[2] == y[2], (Fs, i, hex(x[2]), hex(y[2]))
        assert np.array_equal(x[1], y[1]), (Fs, i, p[0] >> 3, np.nonzero(x[1] != y[1])[0][:6])

@pytest.mark.parametrize("Fs", [8000, 12000, 16000, 24000])
def test_celt_rates(Fs):
    _compare(_stream(2051, 2, 960, 8, bitrate=96000), Fs, 2)
    _compare(_stream(2051, 1, 480, 8, bitrate=48000), Fs, 1, lose=(3,))
    _compare(_stream(2051, 2, 120, 12, bitrate=64000, seed=3), Fs, 1)

@pytest.mark.parametrize("Fs", [8000, 12000, 16000, 24000])
def test_silk_rates(Fs):
    _compare(_stream(2048, 1, 960, 8, bitrate=20000, force_mode=1000, bandwidth=1103), Fs, 1)
    _compare(_stream(2048, 2, 960, 8, bitrate=32000, force_mode=1000, bandwidth=1101, seed=2), Fs, 2, lose=(4,))
    _compare(_stream(2048, 1, 1920, 5, bitrate=16000, force_mode=1000, bandwidth=1102, seed=4), Fs, 2)

@pytest.mark.parametrize("Fs", [8000, 16000, 24000])
def test_hybrid_and_switches(Fs):
    _compare(_stream(2049, 2, 960, 8, bitrate=64000, force_mode=1001, bandwidth=1105), Fs, 2, lose=(5,))
    sched = {3: dict(force_mode=1002), 6: dict(force_mode=1000, bandwidth=1103), 9: dict(force_mode=1001, bandwidth=1104)}
    _compare(_stream(2049, 1, 960, 12, schedule=sched, bitrate=40000, force_mode=1001, bandwidth=1105, seed=7), Fs, 1)

def test_toc_only_packets_change_nothing_but_the_frame_size():
    """A one-byte packet (TOC only: what DTX sends) is concealed in the PREVIOUS mode with the band limit in force -- its own TOC's bandwidth does not apply, because
    opus_decode_frame runs it with bandwidth = 0 (src/opus_decoder.c:316-366, :538) -- and as the FEC source of the previous loss it has no LBRR data either: the call
    conceals and reports a final range of 0 (:676).  Both were found by replaying the decode calls of the reference's tests/test_opus_encode.c (fuzz section) against
    the compiled reference decoder: hybrid SWB -> one-byte hybrid FB packet -> data again."""
    def toc_only(config, stereo): return bytes([(config << 3) | (4 if stereo else 0)])
    pk = _stream(2049, 2, 480, 10, bitrate=48000, force_mode=1001, bandwidth=1104, seed=11)          # hybrid SWB, 10 ms, stereo
    fb = _stream(2049, 2, 480, 4, bitrate=64000, force_mode=1001, bandwidth=1105, seed=12)           # hybrid FB afterwards
    seq = pk[:6] + [toc_only(14, True)] + fb[:2] + [toc_only(12, True), toc_only(31, True)] + pk[6:]
    for Fs, ch in ((48000, 2), (16000, 1)): _compare(seq, Fs, ch)
    sk = _stream(2048, 1, 960, 8, bitrate=20000, force_mode=1000, bandwidth=1103, inband_fec=1, packet_loss=20, seed=13)
    a = capi.Dec("ref", 48000, 2); b = capi.Dec(WHICH, 48000, 2)
    for i, p in enumerate(sk[:5]):
        x = a.decode(p); y = b.decode(p); assert x[0] == y[0] and x[2] == y[2] and np.array_equal(x[1], y[1]), i
    for fec_src in (toc_only(9, False), sk[6]):                                                        # frame lost; the next packet (one-byte / real) as FEC source, then itself
        x = a.decode(fec_src, 960, 1); y = b.decode(fec_src, 960, 1)
        assert x[0] == y[0] and x[2] == y[2] and np.array_equal(x[1], y[1]), (len(fec_src), hex(x[2]), hex(y[2]))
        x = a.decode(fec_src); y = b.decode(fec_src)
        assert x[0] == y[0] and x[2] == y[2] and np.array_equal(x[1], y[1]), len(fec_src)

CORRUPT_HYBRID = bytes.fromhex("6c87fd0b6fe495ccd17a395af820bba695c6bce5cdc3a41ba9247539fccc7bed043755c8bbc01d7efd96c9ce1b5710c04b8b2ea1b423a64b32360322db6b6b74b9d19a3907bf0fc640947fee09a726"
                               "db869a898723eb640341b8c315149227aa38fa309ceab4b988139f6819bb1614344c49f0567a1b2dbac0c4751a0033")

def test_corrupt_redundancy_length_and_losses_longer_than_a_packet():
    """Two more finds of the decode-call replay (tools/replay_decode_trace.py) in the fuzz section of the reference's test_opus_encode: a hybrid packet whose (corrupted)
    redundancy length leaves no payload decodes to a final range of 0 (src/opus_decoder.c:517 `len = 0`, :676), and a loss longer than the longest packet -- here
    360 ms asked of a 16 kHz decoder in one call -- is concealed in full, 20 ms or less at a time (:756-769)."""
    for Fs, ch in ((48000, 2), (16000, 1), (8000, 2)):
        a = capi.Dec("ref", Fs, ch); b = capi.Dec(WHICH, Fs, ch)
        for p in _stream(2049, 2, 960, 3, bitrate=40000, force_mode=1001, bandwidth=1104, seed=21) + [CORRUPT_HYBRID, CORRUPT_HYBRID]:
            x = a.decode(p); y = b.decode(p)
            assert x[0] == y[0] and x[2] == y[2] and np.array_equal(x[1], y[1]), (Fs, ch, hex(x[2]), hex(y[2]))
        assert x[2] == 0
    a = capi.Dec("ref", 16000, 1); b = capi.Dec(WHICH, 16000, 1)
    pk = _stream(2048, 1, 960, 6, bitrate=20000, force_mode=1000, bandwidth=1103, seed=22)
    for p in pk[:4]: a.decode(p); b.decode(p)
    x = a.decode(b"", 5760); y = b.decode(b"", 5760)
    assert x[0] == y[0] == 5760 and np.array_equal(x[1], y[1])
    assert a.get(4039) == b.get(4039) == 5760                       # OPUS_GET_LAST_PACKET_DURATION
    for p in pk[4:]:
        x = a.decode(p); y = b.decode(p)
        assert x[0] == y[0] and x[2] == y[2] and np.array_equal(x[1], y[1])

def test_transition_fade_source_keeps_the_old_band_limit():
    """hybrid fullband -> 2.5 ms narrowband CELT with no redundancy frame: the concealed fade source of the transition runs BEFORE the new packet's band limit is
    applied (src/opus_decoder.c:388-393 and :540-544 come before :547), so it still has the old frame's bands 17..20.  Found by the decode-call replay of test_opus_decode."""
    hyb = _stream(2049, 2, 960, 3, bitrate=64000, force_mode=1001, bandwidth=1105, seed=31)
    nb = _stream(2051, 1, 120, 3, bitrate=32000, bandwidth=1101, seed=32)
    swb = _stream(2049, 1, 480, 3, bitrate=40000, force_mode=1001, bandwidth=1104, seed=33)
    for Fs, ch in ((48000, 2), (48000, 1), (24000, 2)): _compare(hyb + nb + swb + nb + hyb, Fs, ch)
