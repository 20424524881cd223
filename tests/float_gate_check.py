"""tests/float_gate_check.py — the float-mode half of the parity gate of SURVEY 8d, as far as it applies to a bit-exact fixed-point encoder: every packet this
encoder produces must decode with the reference's FLOAT decoder (the build users deploy) to the encoder's final range, and the float decoder's PCM must agree
with the fixed-point decoder's within the noise floor opus_compare tolerates between conforming decoders (here: SNR >= 60 dB on 16-bit PCM, a far tighter bound
than opus_compare's perceptual threshold).  `which` = "emu" | "gpu".

compare_gate() is the decoder half, with the reference's own judge: every committed bitstream (tests/golden/bitstreams/*.bit, reference-encoded over the mode
matrix of tests/test_opus_encode.c) is decoded by THIS decoder and by the reference's FLOAT decoder, at 48 kHz stereo and mono as tests/run_vectors.sh:77-132 does (plus
three of the lower output rates its RATE argument selects), and the reference's opus_compare (src/opus_compare.c compiled in place -> oracle/_ref/opus_compare) must print its PASS verdict for each pair -- the
conformance criterion of RFC 6716 section 6 applied to this decoder against the build users deploy."""
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


def compare_gate(which, tmp, names=None, rates=((48000, 2), (48000, 1), (24000, 2), (16000, 1), (8000, 1))):
    import glob, os, re, struct, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = os.path.join(root, "oracle/_ref/opus_compare")
    assert os.path.exists(tool), "oracle/_ref/opus_compare not built (make -C oracle ref)"
    files = sorted(glob.glob(os.path.join(root, "tests/golden/bitstreams/*.bit")))
    if names: files = [f for f in files if os.path.basename(f)[:-4] in names]
    assert files
    out = {}
    for f in files:
        data = open(f, "rb").read(); pk = []; p = 0
        while p + 8 <= len(data):
            ln, rng = struct.unpack(">II", data[p:p + 8]); p += 8
            pk.append(data[p:p + ln]); p += ln
        ref = capi.Dec("ref_fl", 48000, 2)                                   # the .dec side is always the reference's 48 kHz stereo output (opus_compare.c:232)
        pb = os.path.join(str(tmp), "ref.dec"); np.concatenate([ref.decode(d, 5760)[1] for d in pk]).astype("<i2").tofile(pb)
        refm = capi.Dec("ref_fl", 48000, 1)                                  # and the "m.dec" alternative run_vectors.sh accepts: the reference decoding to mono, as a stereo file
        pm = os.path.join(str(tmp), "refm.dec"); np.repeat(np.concatenate([refm.decode(d, 5760)[1] for d in pk]), 2, axis=1).astype("<i2").tofile(pm)
        for Fs, ch in rates:
            ours = capi.Dec(which, Fs, ch)
            pa = os.path.join(str(tmp), "ours.sw"); np.concatenate([ours.decode(d, Fs // 25 * 3)[1] for d in pk]).astype("<i2").tofile(pa)
            best = None; txts = []
            for dec in (pb, pm):
                r = subprocess.run([tool] + (["-s"] if ch == 2 else []) + ["-r", str(Fs), dec, pa], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
                txt = r.stdout.decode(errors="replace"); txts.append(txt[-200:])
                m = re.search(r"quality metric: ([-0-9.]+) %", txt)
                if r.returncode == 0 and "PASSES" in txt and m: best = max(best or 0.0, float(m.group(1)))
            assert best is not None, (os.path.basename(f), Fs, ch, txts)
            out[(os.path.basename(f)[:-4], Fs, ch)] = best
    return out
