"""Settings fuzz in the manner of the reference's fuzz_encoder_settings (tests/test_opus_encode.c:211), but checked the way that program cannot: every packet and final
range against the compiled reference.  One encoder per seed -- random API rate, channel count and application -- lives through a series of random setting changes
(bitrate incl. OPUS_AUTO / OPUS_BITRATE_MAX, forced channels, VBR / CVBR / CBR, complexity, maximum bandwidth, signal hint, in-band FEC, expected loss, LSB depth,
prediction, DTX, frame duration 2.5-120 ms), each held for about half a second of a signal that moves between loud music, speech, near-silence and digital silence
(so that DTX, the activity logic and the analysis' resets are all exercised).  Even seeds: against the reference with the float API, analysis on; odd seeds: against the
build without it, analysis off.  The reference's own fuzz, traced on the MI355X, found the CELT-only DTX gating that an earlier, narrower version of this test
(tests/test_hostemu_encoder_modes.py::test_settings_fuzz: one application, one rate) had missed.  Here on the wave emulator; tests/test_gpu_classic_api.py runs more
seeds on the MI355X."""
import ctypes, numpy as np, pytest
import capi, signals
from reflib import ref_fx, ref_fxa
from test_kernel_emu_silkdec import speechy
pytestmark = [pytest.mark.skipif(ref_fx() is None or ref_fxa() is None, reason="oracle/_ref not built"), pytest.mark.timeout(900)]   # (a hang is a finding too: opus_pcm_soft_clip on a NaN was one)
WHICH = "emu"
TRPRE = int(__import__("os").environ.get("OPUS_AMD_TEST_TRPRE", "-1"))            # OPUS_AMD_SET_TRANSIENT_PREPASS of every encoder under test: 1 = the transient analysis' lane pre-pass also for the narrow launches of these tests
PIPELINE = int(__import__("os").environ.get("OPUS_AMD_TEST_PIPELINE", "-1"))      # OPUS_AMD_SET_KERNEL_PIPELINE of every encoder under test (include/opus_amd.h): 1 = every 10 / 20 ms call through the front / quantiser / back kernels
LONG = __import__("os").environ.get("OPUS_AMD_LONG_TESTS") == "1"      # the default CPU suite runs the seeds that once found something plus a fresh one or two; OPUS_AMD_LONG_TESTS=1 adds ranges

def _signal(rng, Fs, ch, nsamp):
    n48 = nsamp * (48000 // Fs)
    base = signals.music(n48 // 960 + 2, seed=int(rng.integers(1 << 20))).astype(np.float64)[:n48]
    sp = speechy(n48 // 960 + 2, 2, int(rng.integers(1 << 20)), 960).astype(np.float64)[:n48]
    out = np.zeros((n48, 2)); pos = 0
    while pos < n48:
        seg = int(rng.integers(4800, 48000)); kind = int(rng.integers(0, 6))
        sl = slice(pos, min(n48, pos + seg))
        if kind <= 1: out[sl] = base[sl]
        elif kind == 2: out[sl] = sp[sl]
        elif kind == 3: out[sl] = np.floor(base[sl] / 512)
        elif kind == 4: out[sl] = 0
        else: out[sl] = np.floor(sp[sl] / 64) + rng.integers(-2, 3, out[sl].shape)
        pos += seg
    out[:2880] = 0                                                          # the reference's generate_music starts with 60 ms of silence too
    x = out[::48000 // Fs].astype(np.int16)
    return np.ascontiguousarray(x if ch == 2 else x[:, 0])

def fuzz(seed, changes=10, hold_ms=500, pipeline=None):
    rng = np.random.default_rng(1000 + seed)
    Fs = int(rng.choice([8000, 12000, 16000, 24000, 48000])); ch = int(rng.choice([1, 2])); app = int(rng.choice([2048, 2049, 2051, 2051]))
    analysis = seed % 2 == 0
    a = capi.Enc("ref_fxa" if analysis else "ref", Fs, ch, app); b = capi.Enc(WHICH, Fs, ch, app)
    b.L.opus_encoder_ctl.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    assert b.L.opus_encoder_ctl(b.st, 11900, int(analysis)) == 0
    assert b.L.opus_encoder_ctl(b.st, 11902, PIPELINE if pipeline is None else pipeline) == 0
    assert b.L.opus_encoder_ctl(b.st, 11906, TRPRE) == 0
    sig = _signal(rng, Fs, ch, Fs * (changes * hold_ms + 2000) // 1000); pos = 0
    hist = []
    for j in range(changes):
        ms_x2 = int(rng.choice([5, 10, 20, 40, 40, 40, 80, 120, 160, 200, 240])); fr = ms_x2 * Fs // 2000
        ctl = dict(bitrate=int(rng.choice([6000, 12000, 16000, 24000, 32000, 48000, 64000, 96000, 510000, -1000, -1])), force_channels=min(ch, int(rng.choice([-1000, -1000, 1, 2]))),
                   vbr=int(rng.choice([0, 1, 1])), vbr_constraint=int(rng.choice([0, 1, 1])), complexity=int(rng.choice([0, 2, 4, 5, 7, 8, 9, 10, 10, 10])),
                   max_bandwidth=int(rng.integers(1101, 1106)), signal=int(rng.choice([-1000, -1000, 3001, 3002])), inband_fec=int(rng.choice([0, 0, 1, 2])),
                   packet_loss=int(rng.choice([0, 1, 2, 5, 20])), lsb_depth=int(rng.choice([8, 16, 24])), prediction_disabled=int(rng.choice([0, 0, 1])), dtx=int(rng.choice([0, 1, 1])))
        if ctl["force_channels"] == 0: ctl["force_channels"] = -1000
        for k, v in ctl.items():
            ra, rb = a.set(k, v), b.set(k, v); assert ra == rb == 0, (seed, j, k, v, ra, rb)
        maxb = int(rng.choice([1500, 1500, 1276, 400, 4000]))
        for i in range(max(3, hold_ms * Fs // 1000 // fr)):
            x = sig[pos:pos + fr]; pos += fr
            p, q = a.encode(x, fr, maxb), b.encode(x, fr, maxb)
            hist.append(p[1])
            assert p == q, (seed, (Fs, ch, app), j, i, fr, maxb, p[1], q[1], "%02x %02x" % (p[0][0] if p[0] else 0, q[0][0] if q[0] else 0), ctl, hist[-12:])
    return hist

# seeds 248-423: the ones of a 900-seed sweep that differed when this test was written (CELT-only DTX in multi-frame calls: the gate is the call's, the activity the frame's;
# the loss term of the equivalent rate before the mode is known)
@pytest.mark.parametrize("seed", list(range(12 if LONG else 2)) + [248, 252, 300, 304, 306, 337, 346, 423])
def test_settings_fuzz_against_the_reference(seed): fuzz(seed)


def fuzz_ms(seed, changes=6, hold_ms=300, pipeline=None):
    """the same for multistream encoders: a random layout (mapping family 0 / 1 / 2 / 255), rate and application, random settings through opus_multistream_encoder_ctl"""
    rng = np.random.default_rng(5000 + seed)
    family = int(rng.choice([0, 1, 1, 2, 255])); Fs = int(rng.choice([8000, 12000, 16000, 24000, 48000, 48000])); app = int(rng.choice([2048, 2049, 2049, 2051]))
    nch = int(rng.choice({0: [1, 2], 1: [1, 2, 3, 4, 5, 6, 7, 8], 2: [1, 4, 6, 9, 11], 255: [1, 2, 3, 5, 7]}[family]))
    analysis = seed % 2 == 0
    R = capi._proto(capi.load("ref_fxa" if analysis else "ref")); E = capi._proto(capi.load(WHICH))
    vp, ci = ctypes.c_void_p, ctypes.c_int
    encs = []
    for L in (R, E):
        L.opus_multistream_surround_encoder_create.restype = vp
        L.opus_multistream_surround_encoder_create.argtypes = [ctypes.c_int32, ci, ci, vp, vp, vp, ci, vp]
        s, c, m, err = ci(), ci(), (ctypes.c_ubyte * 256)(), ci()
        e = L.opus_multistream_surround_encoder_create(Fs, nch, family, ctypes.byref(s), ctypes.byref(c), m, app, ctypes.byref(err))
        assert e and err.value == 0, (seed, family, nch, err.value)
        L.opus_multistream_encode.argtypes = [vp, vp, ci, vp, ctypes.c_int32]
        L.opus_multistream_encoder_destroy.argtypes = [vp]; L.opus_multistream_encoder_destroy.restype = None
        encs.append((L, e, (s.value, c.value, bytes(m[:nch]))))
    assert encs[0][2] == encs[1][2]
    def ctl(L, e, req, v): L.opus_multistream_encoder_ctl.argtypes = [vp, ci, ci]; return L.opus_multistream_encoder_ctl(e, req, v)
    def rng_of(L, e):
        v = ctypes.c_uint32(); L.opus_multistream_encoder_ctl.argtypes = [vp, ci, vp]; assert L.opus_multistream_encoder_ctl(e, 4031, ctypes.byref(v)) == 0; return v.value
    assert ctl(E, encs[1][1], 11900, int(analysis)) == 0
    assert ctl(E, encs[1][1], 11902, PIPELINE if pipeline is None else pipeline) == 0
    assert ctl(E, encs[1][1], 11906, TRPRE) == 0
    cols = [_signal(rng, Fs, 1, Fs * (changes * hold_ms + 1500) // 1000) for _ in range(min(nch, 4))]
    sig = np.ascontiguousarray(np.stack([(cols[c % len(cols)] // (1 + c // len(cols))).astype(np.int16) for c in range(nch)], 1))
    pos = 0; cap = 1500 * nch + 4000
    bufs = [(ctypes.c_ubyte * cap)(), (ctypes.c_ubyte * cap)()]
    for j in range(changes):
        ms_x2 = int(rng.choice([5, 10, 20, 40, 40, 40, 80, 120, 160, 240])); fr = ms_x2 * Fs // 2000
        sets = [(4002, int(rng.choice([16000 * nch, 32000 * nch, 64000 * nch, 96000, 510000, -1000, -1]))), (4006, int(rng.choice([0, 1, 1]))), (4020, int(rng.choice([0, 1, 1]))),
                (4010, int(rng.choice([0, 3, 5, 8, 10, 10]))), (4016, int(rng.choice([0, 1]))), (4012, int(rng.choice([0, 0, 1]))), (4014, int(rng.choice([0, 2, 10]))),
                (4036, int(rng.choice([8, 16, 24]))), (4024, int(rng.choice([-1000, -1000, 3001, 3002]))), (4004, int(rng.integers(1101, 1106)))]
        for req, v in sets:
            ra, rb = ctl(R, encs[0][1], req, v), ctl(E, encs[1][1], req, v); assert ra == rb, (seed, j, req, v, ra, rb)
        maxb = int(rng.choice([cap, cap, 400 * encs[0][2][0], 60 * encs[0][2][0]]))
        for i in range(max(2, hold_ms * Fs // 1000 // fr)):
            x = np.ascontiguousarray(sig[pos:pos + fr]).reshape(-1); pos += fr
            out = []
            for (L, e, _), buf in zip(encs, bufs):
                n = L.opus_multistream_encode(e, x.ctypes.data, fr, buf, maxb)
                out.append((n, bytes(buf[:max(n, 0)]), rng_of(L, e) if n > 0 else None))
            assert out[0] == out[1], (seed, (family, nch, Fs, app), j, i, fr, maxb, out[0][0], out[1][0], dict(sets), out[0][1][:120].hex(), out[1][1][:120].hex())
    for L, e, _ in encs: L.opus_multistream_encoder_destroy(e)

# seeds 101, 197: the two of a 400-seed sweep that differed when this test was written (CELT's own energy-mask pointer is dropped by a CELT reset inside a call; a multi-frame
# call that starts during a stereo -> mono transition leaves force_channels at 1 for good)
@pytest.mark.parametrize("seed", list(range(10 if LONG else 2)) + [101, 197])
def test_multistream_settings_fuzz_against_the_reference(seed): fuzz_ms(seed)


def fuzz_sparse(seed, changes=14, hold_ms=350, pipeline=None):
    """a longer-lived variant: every change touches only a random subset of the settings (so that a setting, or a side effect the encoder left behind, survives many
    changes of the others), and the set includes the forced mode, the forced bandwidth and OPUS_RESET_STATE"""
    rng = np.random.default_rng(9000 + seed)
    Fs = int(rng.choice([8000, 12000, 16000, 24000, 48000, 48000])); ch = int(rng.choice([1, 2, 2])); app = int(rng.choice([2048, 2049, 2049, 2051]))
    analysis = seed % 2 == 0
    a = capi.Enc("ref_fxa" if analysis else "ref", Fs, ch, app); b = capi.Enc(WHICH, Fs, ch, app)
    b.L.opus_encoder_ctl.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    assert b.L.opus_encoder_ctl(b.st, 11900, int(analysis)) == 0
    assert b.L.opus_encoder_ctl(b.st, 11902, PIPELINE if pipeline is None else pipeline) == 0
    assert b.L.opus_encoder_ctl(b.st, 11906, TRPRE) == 0
    sig = _signal(rng, Fs, ch, Fs * (changes * hold_ms + 2500) // 1000); pos = 0
    menu = dict(bitrate=[6000, 9000, 12000, 16000, 24000, 32000, 48000, 64000, 96000, 160000, 510000, -1000, -1], force_channels=[-1000, 1, 2], vbr=[0, 1], vbr_constraint=[0, 1],
                complexity=list(range(11)), max_bandwidth=[1101, 1102, 1103, 1104, 1105], bandwidth=[-1000, -1000, 1101, 1102, 1103, 1104, 1105], signal=[-1000, 3001, 3002],
                inband_fec=[0, 1, 2], packet_loss=[0, 1, 5, 15, 40], lsb_depth=[8, 12, 16, 24], prediction_disabled=[0, 1], dtx=[0, 1], force_mode=[-1000, -1000, 1000, 1001, 1002])
    cur = {}; hist = []; fr = Fs // 50
    # seeds >= 1,000,000 with seed % 4 == 3 also change OPUS_SET_EXPERT_FRAME_DURATION (the caller's buffer then holds more than is coded: the look-ahead of the analysis,
    # src/opus_encoder.c:1247, :2662-2690), drawn from a random stream of its own so that the older seeds -- the pinned finds -- replay unchanged
    rng2 = np.random.default_rng(77000 + seed) if seed >= 1000000 and seed % 4 == 3 else None
    for j in range(changes):
        if rng2 is not None and rng2.random() < 0.4:
            v = int(rng2.choice([5000, 5000, 5001, 5002, 5003, 5004, 5005, 5006, 5009])); ra, rb = a.set("expert_frame_duration", v), b.set("expert_frame_duration", v); assert ra == rb == 0
            cur["expert_frame_duration"] = v
        if rng.random() < 0.08:
            for e in (a, b): e.L.opus_encoder_ctl.argtypes = [ctypes.c_void_p, ctypes.c_int]; assert e.L.opus_encoder_ctl(e.st, 4028) == 0
            cur["reset@"] = j
        for k, vals in menu.items():
            if rng.random() < 0.35:
                v = int(rng.choice(vals))
                if k == "force_channels" and v > ch: v = ch
                ra, rb = a.set(k, v), b.set(k, v); assert ra == rb, (seed, j, k, v, ra, rb)
                if ra == 0: cur[k] = v
        if rng.random() < 0.6: fr = int(rng.choice([5, 10, 20, 40, 40, 40, 80, 120, 160, 200, 240])) * Fs // 2000
        maxb = int(rng.choice([1500, 1500, 1276, 250, 4000, 60]))
        for i in range(max(3, hold_ms * Fs // 1000 // fr)):
            x = sig[pos:pos + fr]; pos += fr
            p, q = a.encode(x, fr, maxb), b.encode(x, fr, maxb)
            hist.append("%d:%02x" % (p[1], p[0][0] if p[0] else 0))
            assert p == q, (seed, (Fs, ch, app), j, i, fr, maxb, p[1], q[1], "%02x %02x" % (p[0][0] if p[0] else 0, q[0][0] if q[0] else 0), cur, hist[-10:])

# seeds 102, 134, 172: the three of a 600-seed sweep that differed when this test was written (OPUS_RESET_STATE keeps the reference's silk_mode structure, and with it what
# the last SILK frame before the reset left there: allowBandwidthSwitch & co.)
# seed 5179 (round 4, an 800-job sweep on the round's last day): a CELT -> SILK switch re-initialises the SILK encoder, which clears the channels' input buffers with the rest
# (they had moved out of the channel record that round and were left alone); the first mono frame after stereo ones averages frame_length samples of channel 1's buffer, of which
# its resampler -- still at the old internal rate when rate and channel count switch together -- writes fewer (enc_API.c:318-326)
@pytest.mark.parametrize("seed", list(range(8 if LONG else 1)) + [102, 134, 172, 5179])
def test_sparse_settings_fuzz_against_the_reference(seed): fuzz_sparse(seed)


def fuzz_dec(seed, changes=10, hold_ms=250):
    """decoder fuzz: packets from the reference encoder under the sparse settings fuzz (every mode, bandwidth, frame size, FEC, DTX, multi-frame), decoded at a random output
    rate and channel count by the reference decoder and this one, with random events in between: lost packets (concealment of a random legal length), FEC recovery from the
    next packet, a frame_size argument larger than needed or too small, OPUS_RESET_STATE, a decoder gain, corrupted and truncated packets.  Sample count, PCM and final range
    of every call must agree"""
    rng = np.random.default_rng(13000 + seed)
    Fs = int(rng.choice([8000, 12000, 16000, 24000, 48000, 48000])); ch = int(rng.choice([1, 2])); app = int(rng.choice([2048, 2049, 2049, 2051]))
    e = capi.Enc("ref", Fs, ch, app)
    outFs = int(rng.choice([8000, 12000, 16000, 24000, 48000, 48000])); outch = int(rng.choice([1, 2]))
    a = capi.Dec("ref", outFs, outch); b = capi.Dec(WHICH, outFs, outch)
    sig = _signal(rng, Fs, ch, Fs * (changes * hold_ms + 2500) // 1000); pos = 0
    menu = dict(bitrate=[6000, 9000, 12000, 16000, 24000, 32000, 48000, 64000, 96000, 160000, -1000], force_channels=[-1000, 1, 2], vbr=[0, 1], complexity=[0, 3, 6, 10],
                max_bandwidth=[1101, 1102, 1103, 1104, 1105], inband_fec=[0, 1, 2], packet_loss=[0, 5, 15, 40], dtx=[0, 1], force_mode=[-1000, -1000, 1000, 1001, 1002], signal=[-1000, 3001, 3002])
    fr = Fs // 50; prev = None; k = 0
    def both(pkt, n, fec=0):
        x, y = a.decode(pkt, n, fec), b.decode(pkt, n, fec)
        assert x[0] == y[0] and x[2] == y[2] and np.array_equal(x[1], y[1]), (seed, (Fs, ch, app, outFs, outch), k, len(pkt or b""), n, fec, x[0], y[0], x[2], y[2],
                                                                                 None if x[0] != y[0] or x[0] <= 0 else np.argwhere(x[1] != y[1])[:3].tolist())
    for j in range(changes):
        for name, vals in menu.items():
            if rng.random() < 0.35:
                v = int(rng.choice(vals))
                if name == "force_channels" and v > ch: v = ch
                e.set(name, v)
        if rng.random() < 0.6: fr = int(rng.choice([5, 10, 20, 40, 40, 80, 120, 160, 240])) * Fs // 2000
        for i in range(max(3, hold_ms * Fs // 1000 // fr)):
            pkt = e.encode(sig[pos:pos + fr], fr, int(rng.choice([1500, 1276, 120])))[0]; pos += fr; k += 1
            if not pkt: continue
            n = fr * outFs // Fs                                            # samples this packet holds at the output rate
            ev = rng.random()
            if ev < 0.10:                                                    # lost: conceal, then (sometimes) recover the lost frame from this packet's FEC data instead
                if rng.random() < 0.5: both(b"", n)
                else: both(pkt, n, fec=1)
                both(pkt, n)
            elif ev < 0.14: both(b"", int(rng.choice([outFs // 400, outFs // 100, outFs // 50, outFs * 3 // 50, outFs // 8]))); both(pkt, n)
            elif ev < 0.18: both(pkt, n + int(rng.integers(1, 400)))        # room to spare
            elif ev < 0.21: both(pkt, max(outFs // 400, n // 2)); both(pkt, n)   # too small (OPUS_BUFFER_TOO_SMALL) and again
            elif ev < 0.24:
                bad = bytearray(pkt)
                for _ in range(int(rng.integers(1, 4))): bad[int(rng.integers(0, len(bad)))] ^= 1 << int(rng.integers(0, 8))
                both(bytes(bad), outFs * 3 // 25)
            elif ev < 0.26 and len(pkt) > 3: both(pkt[:int(rng.integers(1, len(pkt)))], outFs * 3 // 25)
            elif ev < 0.28:
                for d in (a, b): d.L.opus_decoder_ctl.argtypes = [ctypes.c_void_p, ctypes.c_int]; assert d.L.opus_decoder_ctl(d.st, 4028) == 0
                both(pkt, n)
            elif ev < 0.30:
                g = int(rng.integers(-3000, 3000)); assert a.set("gain", g) == b.set("gain", g) == 0
                both(pkt, n)
            else: both(pkt, n)

# seeds 6-182: some of the ~70 of a 200-seed sweep that differed when this test was written -- two causes: with OPUS_SET_GAIN the concealed fade source of a mode transition
# carries the gain already when it is mixed in (it comes out of a nested opus_decode_frame), and the reset of the SILK decoder on a CELT -> SILK switch clears the
# comfort-noise excitation buffer with the rest of the state (a stale one is drawn from by the next concealment)
@pytest.mark.parametrize("seed", list(range(10 if LONG else 2)) + [16, 39, 115, 143, 150, 182])
def test_decoder_fuzz_against_the_reference(seed): fuzz_dec(seed)


def fuzz_ms_dec(seed, nframes=40):
    """multistream decoder fuzz: packets of a reference multistream encoder (random layout, settings changing along the way), decoded by both multistream decoders at a
    random output rate with lost packets, FEC recovery, a decoder gain, resets, corrupted packets"""
    rng = np.random.default_rng(17000 + seed)
    family = int(rng.choice([0, 1, 1, 255])); Fs = int(rng.choice([8000, 16000, 24000, 48000, 48000])); app = int(rng.choice([2048, 2049, 2051]))
    nch = int(rng.choice({0: [1, 2], 1: [2, 3, 4, 6, 8], 255: [1, 3, 5]}[family]))
    R = capi._proto(capi.load("ref")); E = capi._proto(capi.load(WHICH)); vp, ci = ctypes.c_void_p, ctypes.c_int
    R.opus_multistream_surround_encoder_create.restype = vp
    R.opus_multistream_surround_encoder_create.argtypes = [ctypes.c_int32, ci, ci, vp, vp, vp, ci, vp]
    s, c, m, err = ci(), ci(), (ctypes.c_ubyte * 256)(), ci()
    enc = R.opus_multistream_surround_encoder_create(Fs, nch, family, ctypes.byref(s), ctypes.byref(c), m, app, ctypes.byref(err)); assert enc and err.value == 0
    R.opus_multistream_encode.argtypes = [vp, vp, ci, vp, ctypes.c_int32]
    outFs = int(rng.choice([8000, 12000, 16000, 24000, 48000, 48000]))
    decs = []
    for L in (R, E):
        L.opus_multistream_decoder_create.restype = vp
        L.opus_multistream_decoder_create.argtypes = [ctypes.c_int32, ci, ci, ci, vp, vp]
        d = L.opus_multistream_decoder_create(outFs, nch, s.value, c.value, m, ctypes.byref(err)); assert d and err.value == 0
        L.opus_multistream_decode.argtypes = [vp, ctypes.c_char_p, ctypes.c_int32, vp, ci, ci]
        L.opus_multistream_decoder_destroy.argtypes = [vp]; L.opus_multistream_decoder_destroy.restype = None
        decs.append((L, d))
    def ectl(req, v): R.opus_multistream_encoder_ctl.argtypes = [vp, ci, ci]; return R.opus_multistream_encoder_ctl(enc, req, v)
    def both(pkt, n, fec=0, k=0):
        out = []
        for L, d in decs:
            o = np.zeros((max(n, 1), nch), np.int16)
            r = L.opus_multistream_decode(d, pkt if pkt else None, len(pkt) if pkt else 0, o.ctypes.data, n, fec)
            v = ctypes.c_uint32(); L.opus_multistream_decoder_ctl.argtypes = [vp, ci, vp]; L.opus_multistream_decoder_ctl(d, 4031, ctypes.byref(v))
            out.append((r, o[:max(r, 0)].tobytes(), v.value))
        assert out[0] == out[1], (seed, (family, nch, Fs, app, outFs), k, len(pkt or b""), n, fec, out[0][0], out[1][0], out[0][2], out[1][2])
    cols = [_signal(rng, Fs, 1, Fs * (nframes * 60 + 1500) // 1000) for _ in range(min(nch, 3))]
    sig = np.ascontiguousarray(np.stack([(cols[q % len(cols)] // (1 + q // len(cols))).astype(np.int16) for q in range(nch)], 1)); pos = 0
    buf = (ctypes.c_ubyte * (1500 * nch + 2000))(); fr = Fs // 50
    for k in range(nframes):
        if k % 8 == 0:
            fr = int(rng.choice([10, 20, 40, 40, 80, 120])) * Fs // 2000
            for req, vals in ((4002, [12000 * nch, 24000 * nch, 64000 * nch, -1000]), (4012, [0, 1]), (4014, [0, 10]), (4016, [0, 1]), (4010, [0, 5, 10]), (4004, [1101, 1103, 1105])): ectl(req, int(rng.choice(vals)))
        x = np.ascontiguousarray(sig[pos:pos + fr]).reshape(-1); pos += fr
        n = R.opus_multistream_encode(enc, x.ctypes.data, fr, buf, len(buf)); assert n > 0
        pkt = bytes(buf[:n]); no = fr * outFs // Fs
        ev = rng.random()
        if ev < 0.12:
            if rng.random() < 0.5: both(b"", no, 0, k)
            else: both(pkt, no, 1, k)
            both(pkt, no, 0, k)
        elif ev < 0.16:
            g = int(rng.integers(-2000, 2000))
            for L, d in decs: L.opus_multistream_decoder_ctl.argtypes = [vp, ci, ci]; assert L.opus_multistream_decoder_ctl(d, 4034, g) == 0
            both(pkt, no, 0, k)
        elif ev < 0.19:
            for L, d in decs: L.opus_multistream_decoder_ctl.argtypes = [vp, ci]; assert L.opus_multistream_decoder_ctl(d, 4028) == 0
            both(pkt, no, 0, k)
        elif ev < 0.23:
            bad = bytearray(pkt); bad[int(rng.integers(0, len(bad)))] ^= 1 << int(rng.integers(0, 8)); both(bytes(bad), outFs * 3 // 25, 0, k)
        elif ev < 0.27: both(pkt, no + int(rng.integers(1, 300)), 0, k)
        else: both(pkt, no, 0, k)
    R.opus_multistream_encoder_destroy.argtypes = [vp]; R.opus_multistream_encoder_destroy(enc)
    for L, d in decs: L.opus_multistream_decoder_destroy(d)

@pytest.mark.parametrize("seed", range(8 if LONG else 2))
def test_multistream_decoder_fuzz_against_the_reference(seed): fuzz_ms_dec(seed)


def fuzz_batch(seed, S=5, changes=8, hold_ms=250, pipeline=None):
    """the batch ABI under the same treatment: S streams of one shape stepped together, settings changed per stream (or for all streams at once) through
    opusgpu_enc_batch_ctl between calls, every stream compared with a reference encoder that was given the same history"""
    rng = np.random.default_rng(21000 + seed)
    Fs = int(rng.choice([8000, 16000, 24000, 48000, 48000])); ch = int(rng.choice([1, 2])); app = int(rng.choice([2048, 2049, 2051, 2051]))
    analysis = seed % 2 == 0
    L = capi.load(WHICH); vp, ci = ctypes.c_void_p, ctypes.c_int
    L.opusgpu_enc_batch_create.restype = vp; L.opusgpu_enc_batch_create.argtypes = [ctypes.c_int32, ctypes.c_int32, ci, ci, ci, vp]
    L.opusgpu_enc_batch_ctl.argtypes = [vp, ctypes.c_int32, ci, ctypes.c_int32]; L.opusgpu_enc_batch_destroy.argtypes = [vp]; L.opusgpu_enc_batch_destroy.restype = None
    L.opusgpu_encode_batch.argtypes = [vp, vp, ci, vp, ctypes.c_int32, ctypes.c_int32, vp, vp]
    err = ci(); b = L.opusgpu_enc_batch_create(S, Fs, ch, app, 0, ctypes.byref(err)); assert b and err.value == 0
    assert L.opusgpu_enc_batch_ctl(b, -1, 11900, int(analysis)) == 0
    assert L.opusgpu_enc_batch_ctl(b, -1, 11902, PIPELINE if pipeline is None else pipeline) == 0
    assert L.opusgpu_enc_batch_ctl(b, -1, 11906, TRPRE) == 0
    refs = [capi.Enc("ref_fxa" if analysis else "ref", Fs, ch, app) for _ in range(S)]
    sigs = [_signal(rng, Fs, ch, Fs * (changes * hold_ms + 2000) // 1000) for _ in range(S)]; pos = 0
    menu = dict(bitrate=[8000, 16000, 32000, 64000, 128000, -1000, -1], force_channels=[-1000, 1, 2], vbr=[0, 1], vbr_constraint=[0, 1], complexity=[0, 4, 8, 10, 10], max_bandwidth=[1101, 1103, 1104, 1105],
                bandwidth=[-1000, -1000, 1101, 1103, 1105], signal=[-1000, 3001, 3002], inband_fec=[0, 1], packet_loss=[0, 5, 20], lsb_depth=[8, 16, 24], prediction_disabled=[0, 1], dtx=[0, 1],
                force_mode=[-1000, -1000, 1000, 1001, 1002])
    fr = Fs // 50
    for j in range(changes):
        for k, vals in menu.items():
            r = rng.random()
            if r < 0.25:                                                   # one stream
                i = int(rng.integers(0, S)); v = int(rng.choice(vals))
                if k == "force_channels" and v > ch: v = ch
                ra = refs[i].set(k, v); rb = L.opusgpu_enc_batch_ctl(b, i, capi.REQ[k], v); assert ra == rb, (seed, j, k, v, i, ra, rb)
            elif r < 0.35:                                                 # all streams
                v = int(rng.choice(vals))
                if k == "force_channels" and v > ch: v = ch
                ras = [e.set(k, v) for e in refs]; rb = L.opusgpu_enc_batch_ctl(b, -1, capi.REQ[k], v); assert all(x == rb for x in ras), (seed, j, k, v, ras, rb)
        if rng.random() < 0.1:
            i = int(rng.integers(0, S)); refs[i].L.opus_encoder_ctl.argtypes = [vp, ci]; assert refs[i].L.opus_encoder_ctl(refs[i].st, 4028) == 0; assert L.opusgpu_enc_batch_ctl(b, i, 4028, 0) == 0
        if rng.random() < 0.6: fr = int(rng.choice([5, 10, 20, 40, 40, 40, 80, 120, 240])) * Fs // 2000
        maxb = int(rng.choice([1500, 1276, 300, 4000]))
        nf = -(-fr * 50 // Fs) if fr > Fs // 50 else 1
        stride = max(1280, (min(maxb, 1276 * 6) + 15) // 16 * 16) if nf == 1 else (maxb + 48 + 15) // 16 * 16
        out = (ctypes.c_ubyte * (stride * S))(); lens = (ctypes.c_int32 * S)(); rngs = (ctypes.c_uint32 * S)()
        for i_ in range(max(2, hold_ms * Fs // 1000 // fr)):
            pcm = np.ascontiguousarray(np.stack([s_[pos:pos + fr].reshape(-1) for s_ in sigs])); pos += fr
            r = L.opusgpu_encode_batch(b, pcm.ctypes.data, fr, out, stride, maxb, lens, rngs); assert r == 0, (seed, j, r)
            for i in range(S):
                p = refs[i].encode(sigs[i][pos - fr:pos], fr, maxb)
                n = lens[i]; q = (bytes(out[i * stride:i * stride + max(n, 0)]), n, rngs[i] if n > 0 else p[2])
                assert (p[0], p[1]) == (q[0], q[1]) and (n <= 0 or p[2] == q[2]), (seed, (Fs, ch, app), j, i_, i, fr, maxb, p[1], q[1], "%02x %02x" % (p[0][0] if p[0] else 0, q[0][0] if q[0] else 0))
    L.opusgpu_enc_batch_destroy(b)

# seeds 5, 118: the two of a 420-seed sweep that differed when this test was written (OPUS_RESET_STATE through the batch ctl took the fields a reset keeps from the host mirror
# instead of the device; a call answered with a 'PLC frame' lost the peak signal energy / stereo-width memory / voice ratio it had already updated)
@pytest.mark.parametrize("seed", list(range(8 if LONG else 1)) + [5, 118])
def test_batch_abi_settings_fuzz_against_the_reference(seed): fuzz_batch(seed)


def test_hybrid_stereo_theta_rdo_folds_the_second_band_like_the_reference():
    """sparse fuzz seed 2169 (the last open case of round 3's campaign): a 48 kHz stereo AUDIO encoder, packet loss 40 %, in-band FEC, OPUS_SET_PREDICTION_DISABLED(1), hard
    CBR 96 kb/s, 60 ms calls, complexity >= 8 -- little enough for CELT's four hybrid bands that partitions without pulses fold from the band below.  An encoder that
    resynthesises (the theta RDO compares reconstructions) needs special_hybrid_folding (celt/bands.c:1575) like the decoder, and again before the second RDO attempt of
    band start + 1; without it 39 bytes of one frame differed (same lengths, same final range of the packet, decodable)"""
    fuzz_sparse(2169)


def fuzz_entry(seed, changes=8, hold_ms=300):
    """the 24-bit and float entry points under changing settings: the codec sees their input rounded to 16 bits, the analysis sees it unrounded (downmix_int24 / downmix_float,
    src/opus_encoder.c:748,:804), so the input carries sub-LSB detail; always against the reference with the float API, the entry point re-drawn per call"""
    rng = np.random.default_rng(25000 + seed)
    Fs = int(rng.choice([16000, 24000, 48000, 48000])); ch = int(rng.choice([1, 2])); app = int(rng.choice([2048, 2049, 2051, 2051]))
    a = capi.Enc("ref_fxa", Fs, ch, app); b = capi.Enc(WHICH, Fs, ch, app)
    b.L.opus_encoder_ctl.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]; assert b.L.opus_encoder_ctl(b.st, 11900, 1) == 0
    vp, ci = ctypes.c_void_p, ctypes.c_int
    for X in (a, b):
        X.L.opus_encode24.argtypes = [vp, vp, ci, vp, ctypes.c_int32]; X.L.opus_encode_float.argtypes = [vp, vp, ci, vp, ctypes.c_int32]
    base = _signal(rng, Fs, ch, Fs * (changes * hold_ms + 2000) // 1000).astype(np.float64)
    fine = base + rng.uniform(-0.5, 0.5, base.shape)                       # what a 24-bit / float capture of the same sound holds below the 16-bit LSB
    menu = dict(bitrate=[12000, 24000, 48000, 96000, -1000], vbr=[0, 1], complexity=[8, 9, 10, 10, 10], max_bandwidth=[1101, 1103, 1104, 1105], signal=[-1000, 3001, 3002], dtx=[0, 1], lsb_depth=[8, 16, 24])
    pos = 0; fr = Fs // 50; out = (ctypes.c_ubyte * 4000)()
    def call(X, kind, x, n, maxb):
        if kind == 0: return X.encode(np.round(x).astype(np.int16), n, maxb)
        if kind == 1: v = np.ascontiguousarray(np.round(x * 256).astype(np.int32)); r = X.L.opus_encode24(X.st, v.ctypes.data, n, out, maxb)
        else: v = np.ascontiguousarray((x / 32768.0).astype(np.float32)); r = X.L.opus_encode_float(X.st, v.ctypes.data, n, out, maxb)
        return bytes(out[:max(r, 0)]), r, X.get(4031) & 0xffffffff
    for j in range(changes):
        for k, vals in menu.items():
            if rng.random() < 0.4:
                v = int(rng.choice(vals)); ra, rb = a.set(k, v), b.set(k, v); assert ra == rb, (seed, j, k, v, ra, rb)
        if rng.random() < 0.6: fr = int(rng.choice([5, 10, 20, 40, 40, 40, 80, 120])) * Fs // 2000
        for i in range(max(3, hold_ms * Fs // 1000 // fr)):
            x = fine[pos:pos + fr]; pos += fr; kind = int(rng.integers(0, 3))
            p, q = call(a, kind, x, fr, 1500), call(b, kind, x, fr, 1500)
            assert p == q, (seed, (Fs, ch, app), j, i, fr, kind, p[1], q[1])

# (when this test was written 27 of its first 60 seeds differed: the 24-bit and float entry points said "24 significant bits" where the build this library reproduces says 16 --
# MAX_ENCODING_DEPTH, celt/arch.h:176 -- which moves the noise floor of the dynamic allocation by eight bits on quiet input)
@pytest.mark.parametrize("seed", [0, 3, 4, 9, 13, 25, 26, 45] if LONG else [0, 25, 26])
def test_entry_point_fuzz_against_the_reference(seed): fuzz_entry(seed)


def fuzz_proj(seed, changes=5, hold_ms=200, pipeline=None):
    """projection (ambisonics with mixing matrices, mapping family 3) encoders and decoders of a random order under changing settings: packets, final ranges and decoded PCM"""
    rng = np.random.default_rng(29000 + seed)
    nch = int(rng.choice([4, 6, 9, 11, 16, 18])); Fs = int(rng.choice([16000, 24000, 48000, 48000])); app = int(rng.choice([2048, 2049, 2051]))
    analysis = seed % 2 == 0
    R = capi.load("ref_fxa" if analysis else "ref"); E = capi.load(WHICH); vp, ci = ctypes.c_void_p, ctypes.c_int
    encs = []; decs = []
    for L in (R, E):
        L.opus_projection_ambisonics_encoder_create.restype = vp
        L.opus_projection_ambisonics_encoder_create.argtypes = [ctypes.c_int32, ci, ci, vp, vp, ci, vp]
        s, c, err = ci(), ci(), ci()
        e = L.opus_projection_ambisonics_encoder_create(Fs, nch, 3, ctypes.byref(s), ctypes.byref(c), app, ctypes.byref(err)); assert e and err.value == 0, (seed, nch, err.value)
        L.opus_projection_encode.argtypes = [vp, vp, ci, vp, ctypes.c_int32]
        L.opus_projection_encoder_ctl.argtypes = [vp, ci, vp]
        size = ctypes.c_int32(); assert L.opus_projection_encoder_ctl(e, 6003, ctypes.byref(size)) == 0
        mat = (ctypes.c_ubyte * size.value)(); L.opus_projection_encoder_ctl.argtypes = [vp, ci, vp, ci]; assert L.opus_projection_encoder_ctl(e, 6005, mat, size.value) == 0
        L.opus_projection_decoder_create.restype = vp; L.opus_projection_decoder_create.argtypes = [ctypes.c_int32, ci, ci, ci, vp, ctypes.c_int32, vp]
        d = L.opus_projection_decoder_create(Fs, nch, s.value, c.value, mat, size.value, ctypes.byref(err)); assert d and err.value == 0
        L.opus_projection_decode.argtypes = [vp, ctypes.c_char_p, ctypes.c_int32, vp, ci, ci]
        encs.append((L, e, (s.value, c.value, bytes(mat)))); decs.append((L, d))
    assert encs[0][2] == encs[1][2]
    def ctl(L, e, req, v): L.opus_projection_encoder_ctl.argtypes = [vp, ci, ci]; return L.opus_projection_encoder_ctl(e, req, v)
    assert ctl(E, encs[1][1], 11900, int(analysis)) == 0
    assert ctl(E, encs[1][1], 11902, PIPELINE if pipeline is None else pipeline) == 0
    assert ctl(E, encs[1][1], 11906, TRPRE) == 0
    cols = [_signal(rng, Fs, 1, Fs * (changes * hold_ms + 1500) // 1000) for _ in range(4)]
    sig = np.ascontiguousarray(np.stack([(cols[q % 4] // (1 + q // 4)).astype(np.int16) for q in range(nch)], 1)); pos = 0
    cap = 1500 * nch; bufs = [(ctypes.c_ubyte * cap)(), (ctypes.c_ubyte * cap)()]
    for j in range(changes):
        fr = int(rng.choice([10, 20, 40, 40, 80])) * Fs // 2000
        for req, vals in ((4002, [16000 * nch, 48000 * nch, -1000, -1]), (4006, [0, 1]), (4010, [0, 5, 10]), (4020, [0, 1])):
            v = int(rng.choice(vals)); ra, rb = ctl(R, encs[0][1], req, v), ctl(E, encs[1][1], req, v); assert ra == rb, (seed, j, req, v, ra, rb)
        for i in range(max(2, hold_ms * Fs // 1000 // fr)):
            x = np.ascontiguousarray(sig[pos:pos + fr]).reshape(-1); pos += fr
            out = []
            for (L, e, _), buf in zip(encs, bufs):
                n = L.opus_projection_encode(e, x.ctypes.data, fr, buf, cap)
                v = ctypes.c_uint32(); L.opus_projection_encoder_ctl.argtypes = [vp, ci, vp]; L.opus_projection_encoder_ctl(e, 4031, ctypes.byref(v))
                out.append((n, bytes(buf[:max(n, 0)]), v.value))
            assert out[0] == out[1], (seed, (nch, Fs, app), j, i, fr, out[0][0], out[1][0])
            pcm = []
            for L, d in decs:
                o = np.zeros((fr, nch), np.int16); r = L.opus_projection_decode(d, out[0][1], out[0][0], o.ctypes.data, fr, 0); pcm.append((r, o.tobytes()))
            assert pcm[0] == pcm[1], (seed, (nch, Fs, app), j, i, "decode", pcm[0][0], pcm[1][0])
    for (L, e, _), (_, d) in zip(encs, decs):
        L.opus_projection_encoder_destroy.argtypes = [vp]; L.opus_projection_encoder_destroy(e); L.opus_projection_decoder_destroy.argtypes = [vp]; L.opus_projection_decoder_destroy(d)

@pytest.mark.parametrize("seed", range(6 if LONG else 1))
def test_projection_fuzz_against_the_reference(seed): fuzz_proj(seed)


def fuzz_packets(seed, rounds=60):
    """the packet toolkit and the repacketizer (src/opus.c, src/repacketizer.c: host code, rewritten from RFC 6716 in round 2) against the reference's: packets of every mode
    and framing from the reference encoder -- whole, truncated, with bytes flipped, and random bytes -- through opus_packet_parse / _get_* / _has_lbrr, pad / unpad (single and
    multistream), and the repacketizer (runs of compatible and incompatible packets, random output ranges and buffer sizes): return codes and bytes"""
    rng = np.random.default_rng(33000 + seed)
    R, E = capi.load("ref"), capi.load(WHICH); vp, ci = ctypes.c_void_p, ctypes.c_int
    Fs = int(rng.choice([8000, 16000, 48000])); ch = int(rng.choice([1, 2])); app = int(rng.choice([2048, 2049, 2051]))
    e = capi.Enc("ref", Fs, ch, app); sig = _signal(rng, Fs, ch, Fs * 30); pos = 0
    pool = []
    for k in range(rounds):
        if k % 6 == 0:
            for name, vals in (("bitrate", [8000, 24000, 64000, 200000]), ("vbr", [0, 1]), ("inband_fec", [0, 1]), ("packet_loss", [0, 20]), ("force_mode", [-1000, 1000, 1001, 1002]), ("dtx", [0, 1])): e.set(name, int(rng.choice(vals)))
        fr = int(rng.choice([5, 10, 20, 40, 40, 80, 120, 240])) * Fs // 2000
        p = e.encode(sig[pos:pos + fr], fr, int(rng.choice([1276, 300, 4000])))[0]; pos += fr
        if p: pool.append(p)
    def variants(p):
        yield p
        if len(p) > 2: yield p[:int(rng.integers(1, len(p)))]
        b = bytearray(p); b[int(rng.integers(0, min(len(b), 4)))] ^= 1 << int(rng.integers(0, 8)); yield bytes(b)
        yield bytes(rng.integers(0, 256, int(rng.integers(1, 40))).astype(np.uint8))
    for L in (R, E):
        L.opus_packet_parse.argtypes = [ctypes.c_char_p, ctypes.c_int32, vp, vp, vp, vp]
        L.opus_packet_get_nb_frames.argtypes = [ctypes.c_char_p, ctypes.c_int32]; L.opus_packet_get_nb_samples.argtypes = [ctypes.c_char_p, ctypes.c_int32, ctypes.c_int32]
        L.opus_packet_has_lbrr.argtypes = [ctypes.c_char_p, ctypes.c_int32]; L.opus_packet_get_bandwidth.argtypes = [ctypes.c_char_p]; L.opus_packet_get_nb_channels.argtypes = [ctypes.c_char_p]
        L.opus_packet_get_samples_per_frame.argtypes = [ctypes.c_char_p, ctypes.c_int32]
        L.opus_packet_pad.argtypes = [vp, ctypes.c_int32, ctypes.c_int32]; L.opus_packet_unpad.argtypes = [vp, ctypes.c_int32]
        L.opus_multistream_packet_pad.argtypes = [vp, ctypes.c_int32, ctypes.c_int32, ci]; L.opus_multistream_packet_unpad.argtypes = [vp, ctypes.c_int32, ci]
        L.opus_repacketizer_create.restype = vp; L.opus_repacketizer_cat.argtypes = [vp, ctypes.c_char_p, ctypes.c_int32]; L.opus_repacketizer_init.argtypes = [vp]; L.opus_repacketizer_init.restype = vp
        L.opus_repacketizer_out_range.argtypes = [vp, ci, ci, vp, ctypes.c_int32]; L.opus_repacketizer_out.argtypes = [vp, vp, ctypes.c_int32]; L.opus_repacketizer_get_nb_frames.argtypes = [vp]
        L.opus_repacketizer_destroy.argtypes = [vp]
    def info(L, p):
        toc = ctypes.c_ubyte(); sizes = (ctypes.c_int16 * 48)(); frames = (vp * 48)(); off = ci()
        buf = ctypes.create_string_buffer(p, len(p))
        n = L.opus_packet_parse(buf, len(p), ctypes.byref(toc), frames, sizes, ctypes.byref(off))
        base = ctypes.addressof(buf)
        fr = [(frames[i] - base, sizes[i]) for i in range(max(n, 0))]
        return (n, toc.value if n > 0 else 0, fr, off.value if n > 0 else 0, L.opus_packet_get_nb_frames(p, len(p)), L.opus_packet_get_nb_samples(p, len(p), 48000), L.opus_packet_has_lbrr(p, len(p)),
                L.opus_packet_get_bandwidth(p), L.opus_packet_get_nb_channels(p), L.opus_packet_get_samples_per_frame(p, 16000))
    for p in pool:
        for q in variants(p):
            a, b = info(R, q), info(E, q); assert a == b, (seed, "parse", q[:6].hex(), len(q), a[:2], b[:2])
            if q is not p: continue                                       # pad / unpad only on well-formed packets: what the reference does with NON-ZERO padding bytes (it reads
                                                                          # them as packet extensions, src/extensions.c, and rejects or re-generates them) is out of scope here
            for new_len in (len(q), len(q) + 1, len(q) + 2, len(q) + int(rng.integers(3, 700)), max(1, len(q) - 1)):
                out = []
                for L in (R, E):
                    buf = (ctypes.c_ubyte * (new_len + 8))(*q[:new_len + 8]); r = L.opus_packet_pad(buf, len(q), new_len)
                    r2 = L.opus_packet_unpad(buf, new_len) if r == 0 else None
                    out.append((r, bytes(buf[:new_len]) if r == 0 else None, r2, bytes(buf[:max(r2 or 0, 0)])))
                assert out[0] == out[1], (seed, "pad", q[:4].hex(), len(q), new_len, out[0][0], out[1][0], out[0][2], out[1][2])
    for _ in range(rounds):                                               # repacketizer: runs of packets
        k = int(rng.integers(1, 6)); run = [pool[int(rng.integers(0, len(pool)))] for _ in range(k)]
        if rng.random() < 0.6: run = [run[0]] * k if rng.random() < 0.3 else [p for p in pool if p[0] & 0xfc == run[0][0] & 0xfc][:k] or run
        res = []; tight = int(rng.integers(1, 400))
        for L in (R, E):
            rp = L.opus_repacketizer_create(); rets = [L.opus_repacketizer_cat(rp, p, len(p)) for p in run]; nb = L.opus_repacketizer_get_nb_frames(rp)
            outs = []
            for (b0, e0, maxlen) in ((0, nb, 8000), (0, max(1, nb // 2), 8000), (nb // 2, nb, 60), (0, nb, tight), (1, 0, 100), (0, nb + 1, 100)):
                buf = (ctypes.c_ubyte * 8000)(); r = L.opus_repacketizer_out_range(rp, b0, e0, buf, maxlen); outs.append((r, bytes(buf[:max(r, 0)])))
            buf = (ctypes.c_ubyte * 8000)(); r = L.opus_repacketizer_out(rp, buf, 8000); outs.append((r, bytes(buf[:max(r, 0)])))
            L.opus_repacketizer_destroy(rp); res.append((rets, nb, outs))
        assert res[0] == res[1], (seed, "repacketizer", [p[:1].hex() for p in run], res[0][0], res[1][0], res[0][1], res[1][1], [o[0] for o in res[0][2]], [o[0] for o in res[1][2]])

@pytest.mark.parametrize("seed", range(6 if LONG else 2))
def test_packet_toolkit_fuzz_against_the_reference(seed): fuzz_packets(seed)


def fuzz_float_out(seed):
    """the float side of the decoder API: opus_decode_float / opus_decode24 (sample conversion of the fixed-point decoder's output) and opus_pcm_soft_clip (src/opus.c:39, float
    arithmetic: compared bit for bit against the reference with the float API) on random and on clipping input, with the memory carried across calls"""
    rng = np.random.default_rng(37000 + seed)
    R, E = capi.load("ref_fxa"), capi.load(WHICH); vp, ci = ctypes.c_void_p, ctypes.c_int
    for L in (R, E): L.opus_pcm_soft_clip.argtypes = [vp, ci, ci, vp]; L.opus_pcm_soft_clip.restype = None
    for ch in (1, 2):
        mem = [np.zeros(2, np.float32), np.zeros(2, np.float32)]
        for k in range(30):
            n = int(rng.choice([1, 7, 120, 480, 960]))
            kind = int(rng.integers(0, 4))
            x = (rng.standard_normal((n, ch)) * (0.3, 0.9, 1.5, 4.0)[kind]).astype(np.float32)
            if kind == 3: x[rng.integers(0, n)] = np.float32(np.nan) if rng.random() < 0.3 else np.float32(3.0)
            out = []
            for i, L in enumerate((R, E)):
                y = np.ascontiguousarray(x.copy()); L.opus_pcm_soft_clip(y.ctypes.data, n, ch, mem[i].ctypes.data); out.append((y.view(np.uint32).tobytes(), mem[i].view(np.uint32).tobytes()))
            assert out[0] == out[1], (seed, "soft_clip", ch, k, n, kind)
    Fs = int(rng.choice([16000, 48000])); ch = int(rng.choice([1, 2]))
    e = capi.Enc("ref", Fs, ch, 2049, bitrate=32000 * ch); sig = _signal(rng, Fs, ch, Fs * 2)
    outFs = int(rng.choice([8000, 24000, 48000])); outch = int(rng.choice([1, 2]))
    decs = [(L, capi.Dec(w, outFs, outch)) for L, w in ((R, "ref_fxa"), (E, WHICH))]
    for L, _ in decs:
        L.opus_decode_float.argtypes = [vp, ctypes.c_char_p, ctypes.c_int32, vp, ci, ci]; L.opus_decode24.argtypes = [vp, ctypes.c_char_p, ctypes.c_int32, vp, ci, ci]
    fr = Fs // 50; no = outFs // 50
    for i in range(40):
        pkt = e.encode(sig[i * fr:(i + 1) * fr], fr)[0]
        if rng.random() < 0.1: pkt = b""
        kind = int(rng.integers(0, 2)); res = []
        for L, d in decs:
            if kind == 0: o = np.zeros((no, outch), np.float32); r = L.opus_decode_float(d.st, pkt if pkt else None, len(pkt), o.ctypes.data, no, 0)
            else: o = np.zeros((no, outch), np.int32); r = L.opus_decode24(d.st, pkt if pkt else None, len(pkt), o.ctypes.data, no, 0)
            res.append((r, o.tobytes()))
        assert res[0] == res[1], (seed, "decode", kind, i, res[0][0], res[1][0])

# (when this test was written opus_pcm_soft_clip never returned on input holding a NaN: its scan for the next sample outside [-1, 1] stopped AT the NaN, the reference's
# predicate walks past it)
@pytest.mark.parametrize("seed", range(4 if LONG else 2))
def test_float_output_fuzz_against_the_reference(seed): fuzz_float_out(seed)


# ---- the same fuzzers with every 10 / 20 ms call forced through the front / pred / quantiser / back kernel pipeline (OPUS_AMD_SET_KERNEL_PIPELINE(4): the pred stage as lane + wave kernels), against the reference ----
# sparse 7770080, batch 7770010: the round-4 review's finds -- the LBRR side stream of the packet before is owed at the head of the first packet after in-band FEC goes 1 -> 0
# (enc_API.c:364-404), and the front kernel coded it into its 64-byte header window; the emulator's LDS watch (hip_stub.h: the window now ends at an inaccessible page, loads
# included) aborts on the spot, on the GPU the stores were dropped and the packet differed from byte 64 on with the same final range
@pytest.mark.parametrize("seed", [7770080, 3000003] + ([7770000, 7770001, 3000007] if LONG else []))       # (3000003, 3000007: with OPUS_SET_EXPERT_FRAME_DURATION changes)
def test_sparse_settings_fuzz_through_the_pipeline(seed): fuzz_sparse(seed, pipeline=4)

@pytest.mark.parametrize("seed", [7770010] + ([7770000, 7770001] if LONG else []))
def test_batch_abi_settings_fuzz_through_the_pipeline(seed): fuzz_batch(seed, pipeline=4)

@pytest.mark.parametrize("seed", [1] + ([248, 300, 7770000] if LONG else []))
def test_settings_fuzz_through_the_pipeline(seed): fuzz(seed, pipeline=4)

@pytest.mark.parametrize("seed", [7770080] + ([1, 3000003] if LONG else []))
def test_sparse_settings_fuzz_through_the_one_wave_pred_kernel(seed): fuzz_sparse(seed, pipeline=3)

@pytest.mark.parametrize("pipeline", [4, 3, 1, 2, 0])
@pytest.mark.parametrize("ch", [1, 2])
def test_pending_lbrr_after_fec_is_switched_off(pipeline, ch):
    """the deterministic case of the round-4 review (BASELINE config-3 shape): 16 kHz VOIP, SILK forced, complexity 10, 96 kb/s, in-band FEC with 20 % expected loss for ten
    frames, then OPUS_SET_INBAND_FEC(0): frame 10 still carries frame 9's LBRR copy in front of its own payload (enc_API.c:364-404)"""
    a = capi.Enc("ref", 16000, ch, 2048); b = capi.Enc(WHICH, 16000, ch, 2048)
    b.L.opus_encoder_ctl.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    assert b.L.opus_encoder_ctl(b.st, 11900, 0) == 0 and b.L.opus_encoder_ctl(b.st, 11902, pipeline) == 0
    g = ctypes.c_int32(); b.L.opus_encoder_ctl.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]; assert b.L.opus_encoder_ctl(b.st, 11903, ctypes.byref(g)) == 0 and g.value == pipeline
    for k, v in dict(force_mode=1000, complexity=10, bitrate=96000 * ch, inband_fec=1, packet_loss=20).items(): assert a.set(k, v) == b.set(k, v) == 0
    x = speechy(20, 2, 77, 960)[::3].astype(np.int16); x = np.ascontiguousarray(x if ch == 2 else x[:, 0])
    for i in range(16):
        if i == 10: assert a.set("inband_fec", 0) == b.set("inband_fec", 0) == 0
        p, q = a.encode(x[i * 320:(i + 1) * 320], 320), b.encode(x[i * 320:(i + 1) * 320], 320)
        assert p == q, (i, p[1], q[1], next((k for k in range(min(len(p[0]), len(q[0]))) if p[0][k] != q[0][k]), None))
