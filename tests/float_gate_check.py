"""tests/float_gate_check.py — the float-mode half of the parity gate of SURVEY 8d, as far as it applies to a bit-exact fixed-point encoder: every packet this
encoder produces must decode with the reference's FLOAT decoder (the build users deploy) to the encoder's final range, and the float decoder's PCM must agree
with the fixed-point decoder's within the noise floor opus_compare tolerates between conforming decoders (here: SNR >= 60 dB on 16-bit PCM, a far tighter bound
than opus_compare's perceptual threshold).  `which` = "emu" | "gpu"."""
import numpy as np
import capi
from test_kernel_emu_silkdec import speechy

CASES = [("config 2", 48000, 2, 2051, dict(bitrate=128000, complexity=10)),
         ("config 3", 16000, 1, 2048, dict(force_mode=1000, bandwidth=1103, bitrate=24000, complexity=10)),
         ("config 4", 48000, 2, 2049, dict(force_mode=1001, bandwidth=1105, bitrate=128000, complexity=10)),
         ("auto 24 kHz", 24000, 1, 2049, dict(bitrate=40000))]

def check(which, name, Fs, ch, app, ctl, frames=12):
    n = Fs // 50
    sig = np.ascontiguousarray(speechy(frames + 1, ch, 321, 960)[::48000 // Fs])
    e = capi.Enc(which, Fs, ch, app, **ctl)
    dfl = capi.Dec("ref_fl", Fs, ch); dfx = capi.Dec("ref", Fs, ch)
    num = den = 0.0
    for f in range(frames):
        pkt, ln, rng = e.encode(np.ascontiguousarray(sig[f * n:(f + 1) * n]), n)
        assert ln > 0, (name, f, ln)
        a = dfl.decode(pkt, n); b = dfx.decode(pkt, n)
        assert a[0] == b[0] == n, (name, f, a[0], b[0])
        assert a[2] == rng == b[2], (name, f, hex(a[2]), hex(rng))          # float decoder, fixed decoder and this encoder end on the same range
        x = a[1].astype(np.float64); y = b[1].astype(np.float64)
        num += float(np.sum(y * y)); den += float(np.sum((x - y) ** 2))
    snr = 10 * np.log10((num + 1e-9) / (den + 1e-9))
    assert snr >= 60.0, (name, snr)
    return snr
