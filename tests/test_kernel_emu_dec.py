"""CPU: the decoder kernel body (opus_amd/csrc/celt_dec_*.h) on the 64-fiber wave emulator against the oracle decoder
(itself pinned to the compiled reference by test_oracle_decoder.py): identical PCM, sample counts and final ranges."""
import ctypes, numpy as np, pytest
import emu_harness as EH, signals
from test_kernel_emu import _build
from test_oracle_encoder import OracleEnc
from test_oracle_decoder import OracleDec
from reflib import oracle

pytestmark = pytest.mark.skipif(oracle() is None, reason="oracle lib not built")

def new_dec_stream(E, channels):
    s = np.zeros(E.emu_sizeof_dec_stream() // 4, np.int32)
    E.emu_dec_stream_reset(s.ctypes.data_as(ctypes.c_void_p), channels)       # the library's own reset (celt_frame.h: oa_dec_stream_reset)
    return s

class EmuDec:
    def __init__(self, channels):
        self.E = _build(); self.ch = channels; self.st = new_dec_stream(self.E, channels)
    def decode(self, pkt, max_frame=5760, fec=0):
        P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        self.E.emu_set_decode_fec(int(fec))
        data = np.frombuffer(pkt + b"\0" * 8, np.uint8).copy()
        lens = np.array([len(pkt)], np.int32); ns = np.zeros(1, np.int32); rg = np.zeros(1, np.uint32)
        pcm = np.zeros((max_frame, self.ch), np.int16)
        self.E.emu_decode_batch(P(self.st), P(data), len(data), P(lens), 1, max_frame, P(pcm), max_frame * self.ch, P(ns), P(rg))
        n = int(ns[0])
        return n, pcm[:max(n, 0)].copy(), int(rg[0])

def _run(enc_ch, dec_ch, sig, frame, nframes, **ctl):
    e = OracleEnc(enc_ch, **ctl); o = OracleDec(dec_ch); k = EmuDec(dec_ch)
    for i in range(nframes):
        pkt, n, erng = e.encode(np.ascontiguousarray(sig[i * frame:(i + 1) * frame]), frame)
        a = o.decode(pkt); b = k.decode(pkt)
        assert a[0] == b[0] == frame, (i, a[0], b[0])
        assert a[2] == b[2] == erng, (i, hex(a[2]), hex(b[2]))
        assert np.array_equal(a[1], b[1]), (i, np.nonzero(a[1] != b[1])[0][:6])

def test_emu_dec_config2():
    _run(2, 2, signals.music(16, seed=0), 960, 16, bitrate=128000, complexity=10)

@pytest.mark.parametrize("channels,bitrate,complexity,frame", [
    (2, 64000, 10, 960), (2, 24000, 10, 960), (2, 510000, 10, 960), (1, 64000, 10, 960), (1, 12000, 5, 960),
    (2, 128000, 10, 480), (2, 128000, 10, 240), (2, 128000, 10, 120), (2, 16000, 10, 960), (2, 8000, 10, 960)])
def test_emu_dec_rates_sizes(channels, bitrate, complexity, frame):
    n = min(12 * 960 // frame, 40)
    _run(channels, channels, signals.music(12, channels=channels, seed=7), frame, n, bitrate=bitrate, complexity=complexity)

@pytest.mark.parametrize("kind", ["bursts", "tone", "silence", "loud"])
def test_emu_dec_signal_kinds(kind):
    sig = {"bursts": signals.noise_bursts(14, seed=2), "tone": signals.tone(14, 440.0), "silence": signals.silence_then_music(14, seed=3),
           "loud": (signals.music(14, seed=4).astype(np.int32) * 4).clip(-32768, 32767).astype(np.int16)}[kind]
    _run(2, 2, sig, 960, 14, bitrate=96000, complexity=10)

def test_emu_dec_channel_mismatch_and_bandwidths():
    _run(1, 2, signals.music(8, channels=1, seed=5), 960, 8, bitrate=48000, complexity=10)
    _run(2, 1, signals.music(8, seed=6), 960, 8, bitrate=96000, complexity=10)
    for bw in (1101, 1103, 1104):
        _run(2, 2, signals.music(6, seed=8), 960, 6, bitrate=64000, complexity=10, bandwidth=bw)

def test_emu_dec_corrupted_packets():
    """garbage in, the reference's garbage out: corrupted packets decode to exactly what the oracle (== reference) produces, state carried"""
    from test_oracle_decoder import _mutations
    rng = np.random.default_rng(21)
    e = OracleEnc(2, bitrate=96000, complexity=5); o = OracleDec(2); k = EmuDec(2)
    sig = signals.music(16, seed=22)
    for i in range(16):
        pkt = e.encode(np.ascontiguousarray(sig[i * 960:(i + 1) * 960]), 960)[0]
        for q in _mutations(pkt, rng, 3) + [pkt]:
            a = o.decode(q); b = k.decode(q)
            assert a[0] == b[0], (i, a[0], b[0], q[:4].hex())
            if a[0] > 0: assert a[2] == b[2] and np.array_equal(a[1], b[1]), (i, q[:4].hex())

@pytest.mark.parametrize("channels,bitrate,frame,pattern", [
    (2, 96000, 960, "single"), (2, 96000, 960, "burst"), (1, 32000, 960, "burst"), (2, 64000, 480, "random"), (2, 128000, 240, "random"),
    (2, 128000, 120, "burst"), (2, 48000, 960, "long"), (1, 64000, 480, "start")])
def test_emu_dec_packet_loss(channels, bitrate, frame, pattern):
    """lost packets (len 0): pitch PLC, fade, noise PLC after long losses, recovery (prefilter_and_fold, energy safety) == oracle == reference"""
    rng = np.random.default_rng(41)
    n = min(40 * 960 // frame, 100)
    sig = signals.music(n * frame // 960 + 1, channels=channels, seed=42)
    e = OracleEnc(channels, bitrate=bitrate, complexity=5); o = OracleDec(channels); k = EmuDec(channels)
    lost = {"single": {10, 20, 30}, "burst": set(range(8, 14)) | set(range(30, 33)), "random": set(np.nonzero(rng.random(n) < 0.2)[0].tolist()),
            "long": set(range(6, 40)), "start": {0, 1, 5}}[pattern]
    for i in range(n):
        pkt = e.encode(np.ascontiguousarray(sig[i * frame:(i + 1) * frame]), frame)[0]
        if i in lost:
            pb = np.zeros((frame, channels), np.int16)
            nb = o.O.oc_opus_decode(o.buf, None, 0, pb.ctypes.data, frame, 0)
            a = k.decode(b"", frame)
            assert nb == a[0] == frame, (i, nb, a[0])
            assert np.array_equal(pb, a[1]), (i, "lost", np.argwhere(pb != a[1])[:4])
        else:
            a = o.decode(pkt); b = k.decode(pkt)
            assert a[0] == b[0] == frame and a[2] == b[2] and np.array_equal(a[1], b[1]), (i, "recv", np.argwhere(a[1] != b[1])[:4])
