"""CPU: SILK-only packets produced by the compiled reference encoder, decoded by the kernel body on the wave emulator and by the compiled
reference decoder: identical PCM and OPUS_GET_FINAL_RANGE packet after packet (state carried)."""
import ctypes, numpy as np, pytest
from reflib import ref_fx
import signals
from test_oracle_encoder import RefEnc
from test_oracle_decoder import RefDec
from test_kernel_emu_dec import EmuDec

pytestmark = pytest.mark.skipif(ref_fx() is None, reason="oracle/_ref not built")

def speechy(nframes, channels=1, seed=0, frame=960):
    """voiced/unvoiced alternation at 48 kHz: glottal-like pulse trains through a resonance + noise bursts"""
    rng = np.random.default_rng(seed); n = nframes * frame
    x = np.zeros(n); t = 0
    while t < n:
        seg = int(rng.integers(2000, 12000))
        if rng.random() < 0.65:
            f0 = rng.uniform(90, 280); period = 48000 / f0; pos = t
            while pos < min(n, t + seg):
                i = int(pos); x[i:i + 40] += np.hanning(80)[40:][:max(0, min(40, n - i))] * rng.uniform(5000, 9000)
                pos += period * rng.uniform(0.98, 1.02)
        else:
            x[t:t + seg] += rng.standard_normal(min(seg, n - t)) * rng.uniform(200, 1500)
        t += seg
    # crude formant colouring
    y = np.zeros(n); a1, a2 = 1.6, -0.8
    for i in range(2, n): y[i] = x[i] + a1 * y[i - 1] * 0.5 + a2 * y[i - 2] * 0.3
    y = np.clip(y * 0.5, -30000, 30000).astype(np.int16)
    return np.stack([y, np.roll(y, 7) // 2 + y // 3], axis=1) if channels == 2 else y.reshape(-1, 1)

def _run(enc_ch, dec_ch, frame, nframes, seed=0, **ctl):
    sig = speechy(nframes, enc_ch, seed, frame)
    e = RefEnc(enc_ch, application=2048, force_mode=1000, **ctl); r = RefDec(dec_ch); k = EmuDec(dec_ch)
    toc_modes = set()
    for i in range(nframes):
        pkt, n, erng = e.encode(np.ascontiguousarray(sig[i * frame:(i + 1) * frame]), frame)
        assert n > 1, (i, n)                                 # no DTX in these streams
        toc_modes.add(pkt[0] >> 7)
        a = r.decode(pkt); b = k.decode(pkt)
        assert a[0] == b[0] == frame, (i, a[0], b[0])
        assert a[2] == b[2] == erng, (i, hex(a[2]), hex(b[2]), hex(erng))
        assert np.array_equal(a[1], b[1]), (i, np.nonzero(a[1] != b[1])[0][:6])
    assert toc_modes == {0}                                  # SILK-only TOCs

@pytest.mark.parametrize("bw,bitrate", [(1103, 24000), (1102, 16000), (1101, 12000), (1103, 40000), (1101, 6000)])
def test_emu_silk_mono(bw, bitrate):
    _run(1, 1, 960, 30, seed=bw + bitrate, bitrate=bitrate, bandwidth=bw)

@pytest.mark.parametrize("frame", [480, 1920, 2880])
def test_emu_silk_frame_sizes(frame):
    _run(1, 1, frame, 15, seed=frame, bitrate=20000, bandwidth=1103)

@pytest.mark.parametrize("bw,bitrate", [(1103, 40000), (1101, 20000)])
def test_emu_silk_stereo(bw, bitrate):
    _run(2, 2, 960, 30, seed=3, bitrate=bitrate, bandwidth=bw)

def test_emu_silk_channel_mismatch():
    _run(1, 2, 960, 12, seed=5, bitrate=20000, bandwidth=1103)       # mono stream, stereo output
    _run(2, 1, 960, 12, seed=6, bitrate=36000, bandwidth=1103)       # stereo stream, mono output

def _run_any(enc_ch, dec_ch, frame, nframes, seed=0, application=2048, schedule=None, expect_modes=None, **ctl):
    """reference encoder with its own mode decisions (or a per-frame ctl schedule) -> reference decoder vs the emulated kernel"""
    sig = speechy(nframes, enc_ch, seed, frame)
    e = RefEnc(enc_ch, application=application, **ctl); r = RefDec(dec_ch); k = EmuDec(dec_ch)
    req = dict(bitrate=4002, bandwidth=4008, max_bandwidth=4004, force_mode=11002, force_channels=4022)
    modes = []
    for i in range(nframes):
        if schedule and i in schedule:
            for kk, v in schedule[i].items(): assert e.L.opus_encoder_ctl(e.st, req[kk], v) == 0
        pkt, n, erng = e.encode(np.ascontiguousarray(sig[i * frame:(i + 1) * frame]), frame)
        assert n > 1, (i, n)
        modes.append("C" if pkt[0] & 0x80 else "H" if (pkt[0] & 0x60) == 0x60 else "S")
        a = r.decode(pkt); b = k.decode(pkt)
        assert a[0] == b[0] == frame, (i, modes[-1], a[0], b[0])
        assert a[2] == b[2] == erng, (i, modes[-1], hex(a[2]), hex(b[2]), hex(erng), "".join(modes))
        assert np.array_equal(a[1], b[1]), (i, modes[-1], "".join(modes), np.nonzero(a[1] != b[1])[0][:6])
    if expect_modes: assert set(modes) >= set(expect_modes), "".join(modes)
    return "".join(modes)

@pytest.mark.parametrize("ch,bitrate,bw,frame", [(1, 32000, 1105, 960), (1, 24000, 1104, 960), (2, 48000, 1105, 960), (1, 28000, 1105, 480), (2, 40000, 1104, 480)])
def test_emu_hybrid(ch, bitrate, bw, frame):
    _run_any(ch, ch, frame, 24, seed=ch + bitrate, force_mode=1001, bitrate=bitrate, bandwidth=bw, expect_modes="H")

def test_emu_silk_hybrid_switching():
    """bandwidth moves between WB (SILK-only) and FB (hybrid) inside one stream: the CELT layer is reset / faded on a silence frame"""
    sched = {0: dict(force_mode=1000, bandwidth=1103), 6: dict(force_mode=1001, bandwidth=1105), 12: dict(force_mode=1000, bandwidth=1103), 18: dict(force_mode=1001, bandwidth=1104)}
    _run_any(1, 1, 960, 24, seed=9, bitrate=28000, schedule=sched, expect_modes="SH")
    _run_any(2, 2, 960, 24, seed=10, bitrate=44000, schedule=sched, expect_modes="SH")

def test_emu_celt_to_silk_transitions():
    """CELT-only -> SILK/hybrid switches: redundancy frames and CELT-concealment cross-fades (the opposite direction needs SILK concealment)"""
    sched = {0: dict(force_mode=1002), 8: dict(force_mode=1000, bandwidth=1103)}
    _run_any(1, 1, 960, 16, seed=11, bitrate=32000, schedule=sched, expect_modes="CS")
    sched = {0: dict(force_mode=1002), 8: dict(force_mode=1001, bandwidth=1105)}
    _run_any(2, 2, 960, 16, seed=12, bitrate=48000, schedule=sched, expect_modes="CH")

def _run_loss(ch, frame, nframes, seed, lose, schedule=None, **ctl):
    """packets listed in `lose` never reach the decoders (opus_decode(NULL, 0, frame)): SILK / hybrid concealment, comfort noise, the energy glue
    on recovery, and concealment-driven mode transitions must match the reference sample for sample"""
    sig = speechy(nframes, ch, seed, frame)
    e = RefEnc(ch, application=2048, **ctl); r = RefDec(ch); k = EmuDec(ch)
    req = dict(bitrate=4002, bandwidth=4008, max_bandwidth=4004, force_mode=11002)
    for i in range(nframes):
        if schedule and i in schedule:
            for kk, v in schedule[i].items(): assert e.L.opus_encoder_ctl(e.st, req[kk], v) == 0
        pkt, n, erng = e.encode(np.ascontiguousarray(sig[i * frame:(i + 1) * frame]), frame)
        if i in lose: pkt = b""
        a = r.decode(pkt, frame); b = k.decode(pkt, frame)
        assert a[0] == b[0] == frame, (i, a[0], b[0])
        assert a[2] == b[2], (i, hex(a[2]), hex(b[2]))
        assert np.array_equal(a[1], b[1]), (i, i in lose, np.nonzero(a[1] != b[1])[0][:6])

@pytest.mark.parametrize("ch,mode,bw,bitrate,frame", [(1, 1000, 1103, 20000, 960), (1, 1000, 1101, 10000, 960), (2, 1000, 1103, 36000, 960), (1, 1001, 1105, 32000, 960),
                                                    (2, 1001, 1104, 44000, 960), (1, 1000, 1102, 16000, 480), (1, 1001, 1105, 30000, 480), (1, 1000, 1103, 20000, 1920)])
def test_emu_silk_packet_loss(ch, mode, bw, bitrate, frame):
    lose = {3, 7, 8, 9, 14, 15, 16, 17, 18, 19, 20, 24}
    _run_loss(ch, frame, 30, seed=mode + bw + frame, lose=lose, force_mode=mode, bandwidth=bw, bitrate=bitrate)

def test_emu_silk_to_celt_transition_and_loss_across_modes():
    sched = {0: dict(force_mode=1000, bandwidth=1103), 8: dict(force_mode=1002, bandwidth=1105), 16: dict(force_mode=1001, bandwidth=1105), 24: dict(force_mode=1002)}
    _run_any(1, 1, 960, 32, seed=21, bitrate=36000, schedule=sched, expect_modes="SCH")
    _run_any(2, 2, 960, 32, seed=22, bitrate=56000, schedule=sched, expect_modes="SCH")
    _run_loss(1, 960, 32, seed=23, lose={7, 8, 15, 16, 23, 24, 25}, schedule=sched, bitrate=36000)     # the packets around every switch are lost

def _run_fec(ch, frame, nframes, seed, lose, **ctl):
    """in-band FEC: every lost packet is recovered from the LBRR copy carried by the NEXT packet (opus_decode(next, decode_fec=1) then
    opus_decode(next, decode_fec=0)), exactly as a jitter buffer drives the reference (src/opus_demo.c:1107-1160)"""
    sig = speechy(nframes, ch, seed, frame)
    e = RefEnc(ch, application=2048, inband_fec=1, packet_loss=25, **ctl); r = RefDec(ch); k = EmuDec(ch)
    pk = [e.encode(np.ascontiguousarray(sig[i * frame:(i + 1) * frame]), frame)[0] for i in range(nframes)]
    used_lbrr = 0
    for i in range(nframes):
        if i in lose:
            if i + 1 < nframes and (i + 1) not in lose:
                a = r.decode(pk[i + 1], frame, fec=1); b = k.decode(pk[i + 1], frame, fec=1); used_lbrr += 1
            else:
                a = r.decode(b"", frame); b = k.decode(b"", frame)
        else:
            a = r.decode(pk[i], frame); b = k.decode(pk[i], frame)
        assert a[0] == b[0] == frame, (i, a[0], b[0])
        assert a[2] == b[2], (i, hex(a[2]), hex(b[2]))
        assert np.array_equal(a[1], b[1]), (i, i in lose, np.nonzero(a[1] != b[1])[0][:6])
    assert used_lbrr > 0

@pytest.mark.parametrize("ch,mode,bw,bitrate,frame", [(1, 1000, 1103, 24000, 960), (2, 1000, 1103, 40000, 960), (1, 1001, 1105, 36000, 960), (1, 1000, 1101, 14000, 1920),
                                                    (1, 1000, 1103, 24000, 480)])
def test_emu_inband_fec(ch, mode, bw, bitrate, frame):
    _run_fec(ch, frame, 26, seed=mode + bw + frame + ch, lose={3, 7, 8, 12, 16, 17, 18, 22}, force_mode=mode, bandwidth=bw, bitrate=bitrate)

def test_emu_fec_packet_changes_parameters():
    """the packet whose LBRR copy is used differs from the previous packet in bandwidth (hybrid SWB -> FB end band), stream channels and frame size, and the request is
    longer than the packet: the leading samples are concealed with the PREVIOUS parameters, the packet's own apply from the LBRR frame on (src/opus_decoder.c:798-823)"""
    ch = 2
    sig = speechy(40, ch, 77, 960)
    e = RefEnc(ch, application=2049, inband_fec=1, packet_loss=25, force_mode=1001, bandwidth=1104, bitrate=40000); r = RefDec(ch); k = EmuDec(ch)
    req = dict(bandwidth=4008, force_channels=4022, bitrate=4002)
    sched = {6: dict(bandwidth=1105), 12: dict(force_channels=1), 18: dict(bandwidth=1104, force_channels=2), 24: dict(bandwidth=1105, bitrate=56000)}
    pos = 0; pk = []
    for i in range(30):
        if i in sched:
            for kk, v in sched[i].items(): assert e.L.opus_encoder_ctl(e.st, req[kk], v) == 0
        fr = 480 if 18 <= i < 24 else 960
        pk.append((e.encode(np.ascontiguousarray(sig[pos:pos + fr]), fr)[0], fr)); pos += fr
    lose = {5, 11, 17, 23}                         # each loss is followed by the first packet with the new parameters
    used = 0
    for i in range(30):
        if i in lose:
            want = pk[i][1]                        # the duration of the lost packet (as a jitter buffer would ask), with the NEXT packet's data and decode_fec = 1
            a = r.decode(pk[i + 1][0], want, fec=1); b = k.decode(pk[i + 1][0], want, fec=1); used += 1
        else:
            a = r.decode(pk[i][0], pk[i][1]); b = k.decode(pk[i][0], pk[i][1])
        assert a[0] == b[0], (i, a[0], b[0])
        assert a[2] == b[2], (i, hex(a[2]), hex(b[2]))
        assert np.array_equal(a[1], b[1]), (i, i in lose, np.nonzero(a[1] != b[1])[0][:6])
    assert used == 4

def test_emu_fec_request_on_celt_packets_conceals():
    """decode_fec on CELT-only packets (no LBRR exists) = concealment (src/opus_decoder.c:791-797)"""
    sig = speechy(10, 1, 3, 960)
    e = RefEnc(1, application=2051, bitrate=48000); r = RefDec(1); k = EmuDec(1)
    for i in range(10):
        pkt = e.encode(np.ascontiguousarray(sig[i * 960:(i + 1) * 960]), 960)[0]
        fec = i in (4, 7)
        a = r.decode(pkt, 960, fec=fec); b = k.decode(pkt, 960, fec=fec)
        assert a[0] == b[0] == 960 and a[2] == b[2] and np.array_equal(a[1], b[1]), i
