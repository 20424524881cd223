"""CPU: the classic encoder API of the product's C ABI on the wave emulator against the compiled reference encoder, packet bytes and final range frame
by frame, for what opus_encode_native / opus_encode_frame_native add around the codecs (src/opus_encoder.c:1182-2657): mode switches in every direction with
their CELT redundancy frames and SILK / CELT prefills, SILK bandwidth switches, calls above 20 ms re-framed as multi-frame packets (40-120 ms), API rates below
48 kHz (CELT zero-stuffing), hard CBR padding, settings changed mid-stream (the reference's tests/test_opus_encode.c:211 fuzz_encoder_settings pattern)."""
import numpy as np, pytest
import capi, signals
from reflib import ref_fx
from test_kernel_emu_silkdec import speechy
pytestmark = pytest.mark.skipif(ref_fx() is None, reason="oracle/_ref not built")
WHICH = "emu"

def sig_for(Fs, ch, nsamp, seed):
    s = speechy((nsamp * 48000 // Fs) // 960 + 2, ch, seed, 960)
    step = 48000 // Fs
    return np.ascontiguousarray(s[::step][:nsamp])

def run(Fs, ch, app, frames, schedule=None, seed=0, maxb=1276, **ctl):
    """frames: list of frame sizes (samples at Fs) per call; schedule: {call index: {ctl: value}} applied to both encoders"""
    a = capi.Enc("ref", Fs, ch, app, **ctl); b = capi.Enc(WHICH, Fs, ch, app, **ctl)
    sig = sig_for(Fs, ch, sum(frames) + 16, seed)
    pos = 0; modes = []
    for i, fr in enumerate(frames):
        if schedule and i in schedule:
            for k, v in schedule[i].items():
                ra = a.set(k, v); rb = b.set(k, v)
                assert ra == rb, (i, k, v, ra, rb)
        pcm = sig[pos:pos + fr]; pos += fr
        mb = maxb[i] if isinstance(maxb, (list, tuple)) else maxb
        x = a.encode(pcm, fr, mb); y = b.encode(pcm, fr, mb)
        modes.append("?" if x[1] <= 0 else "C" if x[0][0] & 0x80 else "H" if (x[0][0] & 0x60) == 0x60 else "S")
        assert x[1] == y[1], (i, fr, "".join(modes), x[1], y[1])
        assert x[2] == y[2], (i, fr, "".join(modes), hex(x[2]), hex(y[2]))
        assert x[0] == y[0], (i, fr, "".join(modes), [k for k in range(len(x[0])) if x[0][k] != y[0][k]][:8])
    return "".join(modes)

def test_silk_celt_switches_mono():
    sched = {4: dict(force_mode=1002), 8: dict(force_mode=1000), 12: dict(force_mode=1002), 14: dict(force_mode=1001), 18: dict(force_mode=1002), 20: dict(force_mode=1000)}
    m = run(48000, 1, 2049, [960] * 24, sched, bitrate=32000, force_mode=1000, bandwidth=1103)
    assert "S" in m and "C" in m, m

def test_hybrid_celt_switches_stereo():
    sched = {3: dict(force_mode=1002), 6: dict(force_mode=1001), 9: dict(force_mode=1002, bitrate=96000), 12: dict(force_mode=1001, bitrate=48000)}
    m = run(48000, 2, 2049, [960] * 16, sched, seed=3, bitrate=64000, force_mode=1001, bandwidth=1105)
    assert "H" in m and "C" in m, m

def test_switches_10ms_and_short_frames():
    sched = {3: dict(force_mode=1002), 7: dict(force_mode=1000), 10: dict(force_mode=1002)}
    run(48000, 1, 2048, [480] * 6 + [240, 120, 480, 480, 480, 240, 480, 480], sched, seed=5, bitrate=28000, force_mode=1000, bandwidth=1103)

def test_auto_mode_rate_sweep():
    sched = {i: dict(bitrate=br) for i, br in zip(range(2, 40, 3), [12000, 48000, 20000, 64000, 9000, 80000, 16000, 40000, 24000, 96000, 14000, 56000, 30000])}
    m = run(48000, 2, 2049, [960] * 40, sched, seed=9, bitrate=24000)
    assert len(set(m)) >= 2, m

def test_silk_bandwidth_switch():
    sched = {5: dict(max_bandwidth=1101), 12: dict(max_bandwidth=1103), 18: dict(bitrate=9000), 26: dict(bitrate=30000)}
    run(16000, 1, 2048, [320] * 34, sched, seed=11, bitrate=24000)

@pytest.mark.parametrize("ms", [40, 60, 80, 100, 120])
def test_long_frames_celt_and_hybrid(ms):
    fr = 48 * ms
    run(48000, 2, 2051, [fr] * 3, seed=ms, bitrate=96000)
    run(48000, 1, 2049, [fr] * 3, seed=ms + 1, bitrate=40000, force_mode=1001, bandwidth=1105)

@pytest.mark.parametrize("ms", [80, 100, 120])
def test_long_frames_silk(ms):
    run(16000, 1, 2048, [16 * ms] * 3, seed=ms, bitrate=20000, force_mode=1000)
    run(48000, 2, 2048, [48 * ms] * 2, seed=ms + 7, bitrate=36000, force_mode=1000, bandwidth=1103, vbr=0)

@pytest.mark.parametrize("Fs", [8000, 12000, 16000, 24000])
def test_celt_below_48k(Fs):
    run(Fs, 2, 2051, [Fs // 50] * 6 + [Fs // 100] * 3 + [Fs // 200, Fs // 400] * 2, seed=Fs // 1000, bitrate=64000)
    run(Fs, 1, 2049, [Fs // 50] * 8, {3: dict(force_mode=1000), 6: dict(force_mode=1002)}, seed=Fs // 1000 + 1, bitrate=32000, force_mode=1002)

def test_cbr_padding_and_tiny_buffers():
    run(48000, 1, 2049, [960] * 8, {3: dict(force_mode=1002), 6: dict(force_mode=1000)}, seed=21, bitrate=24000, vbr=0, force_mode=1000, bandwidth=1103)
    run(48000, 2, 2049, [1920] * 4, seed=22, bitrate=64000, vbr=0)
    run(48000, 1, 2048, [960] * 6, seed=23, bitrate=16000, maxb=[2, 1, 40, 3, 100, 7])

def test_settings_fuzz():
    rng = np.random.default_rng(5)
    Fs = 48000
    frames = [int(rng.choice([120, 240, 480, 960, 960, 960, 1920, 2880])) for _ in range(40)]
    sched = {}
    for i in range(1, 40, 2):
        sched[i] = dict(bitrate=int(rng.integers(6000, 140000)), complexity=int(rng.integers(0, 11)), vbr=int(rng.integers(0, 2)), vbr_constraint=int(rng.integers(0, 2)),
                        force_channels=int(rng.choice([-1000, 1, 2])), max_bandwidth=int(rng.integers(1101, 1106)), signal=int(rng.choice([-1000, 3001, 3002])),
                        inband_fec=int(rng.integers(0, 3)), packet_loss=int(rng.integers(0, 30)), dtx=int(rng.integers(0, 2)), prediction_disabled=int(rng.integers(0, 2)))
    run(Fs, 2, 2049, frames, sched, seed=31)

@pytest.mark.parametrize("ch", [1, 2])
def test_complexity_steps_midstream(ch):
    """every row of the SILK complexity table (silk/control_codec.c:307) entered and left mid-stream in hybrid frames: the noise-shaping order moves 12 -> 24 -> 20 -> 24 ...,
    and the taps above the current order have to survive untouched until the order rises again (found by the parity soak's control schedule)"""
    sched = {i * 4: dict(complexity=c) for i, c in enumerate([6, 10, 2, 8, 0, 7, 4, 10, 1, 6, 9])}
    m = run(48000, ch, 2049, [960] * 46, sched, seed=5, bitrate=160000, force_mode=1001, bandwidth=1105)
    assert set(m) == {"H"}, m
