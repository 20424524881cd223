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
