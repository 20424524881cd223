"""GPU (MI355X): the HIP path, called through the C ABI, must reproduce the reference bit for bit.
Checkers: the plain-C oracle (oracle/libcelt_oracle.so) and, where it travelled with the snapshot, the compiled
reference itself (oracle/_ref/libopus_ref_fx.so).  Bit-exact: packets, lengths, OPUS_GET_FINAL_RANGE."""
import ctypes, os, numpy as np, pytest
import signals
from reflib import oracle, ref_fx

pytestmark = pytest.mark.gpu

def _oa():
    import opus_amd
    return opus_amd

def _check_streams(S, frames, channels, frame, ctl, checker="oracle"):
    oa = _oa()
    from test_oracle_encoder import OracleEnc, RefEnc
    kw = dict(ctl)
    b = oa.EncoderBatch(S, channels=channels)
    req = dict(bitrate=oa.OPUS_SET_BITRATE_REQUEST, complexity=oa.OPUS_SET_COMPLEXITY_REQUEST, vbr_constraint=oa.OPUS_SET_VBR_CONSTRAINT_REQUEST,
               force_channels=oa.OPUS_SET_FORCE_CHANNELS_REQUEST, user_bandwidth=oa.OPUS_SET_BANDWIDTH_REQUEST, max_bandwidth=oa.OPUS_SET_MAX_BANDWIDTH_REQUEST,
               disable_inv=oa.OPUS_SET_PHASE_INVERSION_DISABLED_REQUEST)
    for k, v in kw.items(): b.ctl(req[k], v)
    sigs = [signals.music(frames * frame // 960 + 1, channels=channels, seed=100 + s) if s % 3 else signals.noise_bursts(frames * frame // 960 + 1, channels=channels, seed=s) for s in range(S)]
    Enc = OracleEnc if checker == "oracle" else RefEnc
    if checker == "ref":
        kw = {("bandwidth" if k == "user_bandwidth" else "phase_inv_disabled" if k == "disable_inv" else k): v for k, v in kw.items()}
    chk = [Enc(channels, **kw) for _ in range(S)]
    for i in range(frames):
        pcm = np.stack([np.ascontiguousarray(sigs[s][i * frame:(i + 1) * frame]).reshape(-1) for s in range(S)])
        pk, lens, rng = b.encode(pcm, frame)
        for s in range(S):
            a = chk[s].encode(np.ascontiguousarray(sigs[s][i * frame:(i + 1) * frame]), frame)
            assert (a[0], a[1], a[2]) == (pk[s], int(lens[s]), int(rng[s])), (i, s, a[1], int(lens[s]), hex(a[2]), hex(int(rng[s])))
    b.close()

def test_gpu_config2_vs_oracle():
    """BASELINE config 2 shape: restricted-lowdelay 48 kHz stereo 20 ms, 128 kb/s CVBR, complexity 10; 48 streams x 60 frame-steps."""
    _check_streams(48, 60, 2, 960, dict(bitrate=128000, complexity=10))

@pytest.mark.skipif(ref_fx() is None, reason="compiled reference did not travel")
def test_gpu_config2_vs_reference():
    _check_streams(16, 100, 2, 960, dict(bitrate=128000, complexity=10), checker="ref")

@pytest.mark.parametrize("channels,bitrate,complexity,frame", [
    (2, 64000, 10, 960), (2, 24000, 10, 960), (2, 510000, 10, 960), (1, 64000, 10, 960), (1, 12000, 5, 960),
    (2, 96000, 5, 960), (2, 96000, 0, 960), (2, 128000, 10, 480), (2, 128000, 10, 240), (2, 128000, 10, 120), (2, 8000, 10, 960)])
def test_gpu_rates_sizes(channels, bitrate, complexity, frame):
    _check_streams(8, 30 * 960 // frame if frame >= 480 else 60, channels, frame, dict(bitrate=bitrate, complexity=complexity))

def test_gpu_ctls():
    _check_streams(4, 30, 2, 960, dict(bitrate=96000, complexity=10, vbr_constraint=0))
    _check_streams(4, 30, 2, 960, dict(bitrate=96000, complexity=10, force_channels=1))
    _check_streams(4, 30, 2, 960, dict(bitrate=64000, complexity=10, user_bandwidth=1103))
    _check_streams(4, 30, 2, 960, dict(bitrate=64000, complexity=10, max_bandwidth=1104, disable_inv=1))

def test_gpu_edge_inputs():
    """silence (digital zero), full-scale square, tiny buffers (PLC frame), tone."""
    oa = _oa()
    from test_oracle_encoder import OracleEnc
    cases = [np.zeros((960, 2), np.int16), np.full((960, 2), 32767, np.int16), signals.tone(1, freq=440.0), (signals.music(1) // 4096 * 4096).astype(np.int16)]
    b = oa.EncoderBatch(len(cases), channels=2); b.ctl(oa.OPUS_SET_BITRATE_REQUEST, 128000); b.ctl(oa.OPUS_SET_COMPLEXITY_REQUEST, 10)
    chk = [OracleEnc(2, bitrate=128000, complexity=10) for _ in cases]
    for rep in range(6):
        pcm = np.stack([c.reshape(-1) for c in cases])
        pk, lens, rng = b.encode(pcm, 960)
        for s, c in enumerate(cases):
            a = chk[s].encode(np.ascontiguousarray(c), 960)
            assert (a[0], a[2]) == (pk[s], int(rng[s])), (rep, s)
    for maxb in (2, 3, 10, 60):
        pk, lens, rng = b.encode(pcm, 960, max_data_bytes=maxb)
        for s, c in enumerate(cases):
            a = chk[s].encode(np.ascontiguousarray(c), 960, maxb)
            assert (a[0], a[2]) == (pk[s], int(rng[s])), (maxb, s, a[1], int(lens[s]))
    b.close()

def test_gpu_classic_api_and_state_contract():
    """opus_encoder_create/opus_encode/opus_encoder_ctl drop-in + the memcpy contract (export/import mid-stream)."""
    oa = _oa()
    from test_oracle_encoder import OracleEnc
    sig = signals.music(30, seed=9)
    e = oa.OpusEncoder(48000, 2, oa.OPUS_APPLICATION_RESTRICTED_LOWDELAY)
    e.ctl(oa.OPUS_SET_BITRATE_REQUEST, 128000); e.ctl(oa.OPUS_SET_COMPLEXITY_REQUEST, 10)
    o = OracleEnc(2, bitrate=128000, complexity=10)
    for i in range(12):
        pcm = np.ascontiguousarray(sig[i * 960:(i + 1) * 960])
        a = o.encode(pcm, 960); p = e.encode(pcm, 960)
        assert a[0] == p and a[2] == e.final_range()
    assert oa.OpusEncoder(48000, 2, oa.OPUS_APPLICATION_AUDIO) is not None                       # the SILK-capable encoder (tests/test_gpu_silkenc.py)
    with pytest.raises(oa.OpusError): oa.OpusEncoder(44100, 2, oa.OPUS_APPLICATION_RESTRICTED_LOWDELAY)  # OPUS_BAD_ARG like the reference
    with pytest.raises(oa.OpusError): e.ctl(oa.OPUS_SET_COMPLEXITY_REQUEST, 11)
    # state migrates between batch slots through a flat blob
    b = oa.EncoderBatch(2, channels=2); b.ctl(oa.OPUS_SET_BITRATE_REQUEST, 128000); b.ctl(oa.OPUS_SET_COMPLEXITY_REQUEST, 10)
    o2 = OracleEnc(2, bitrate=128000, complexity=10)
    for i in range(5):
        pcm = np.ascontiguousarray(sig[i * 960:(i + 1) * 960]); o2.encode(pcm, 960)
        b.encode(np.stack([pcm.reshape(-1), pcm.reshape(-1)]), 960)
    blob = b.export_state(0)
    b.reset(); b.import_state(1, blob)
    for i in range(5, 10):
        pcm = np.ascontiguousarray(sig[i * 960:(i + 1) * 960]); a = o2.encode(pcm, 960)
        pk, lens, rng = b.encode(np.stack([np.zeros(1920, np.int16), pcm.reshape(-1)]), 960)
        assert a[0] == pk[1] and a[2] == int(rng[1])
    b.close()

@pytest.mark.parametrize("channels,bitrate,frame", [(2, 128000, 960), (2, 64000, 480), (1, 32000, 960), (2, 6000, 960), (1, 500, 960), (2, 510000, 960)])
def test_gpu_hard_cbr(channels, bitrate, frame):
    """OPUS_SET_VBR(0): constant packet size, code-3 padding where the coder comes up short, TOC-only packets below the useful minimum"""
    oa = _oa()
    from test_oracle_encoder import OracleEnc
    S = 6
    b = oa.EncoderBatch(S, channels=channels)
    b.ctl(oa.OPUS_SET_BITRATE_REQUEST, bitrate); b.ctl(oa.OPUS_SET_COMPLEXITY_REQUEST, 10); b.ctl(oa.OPUS_SET_VBR_REQUEST, 0)
    chk = [OracleEnc(channels, bitrate=bitrate, complexity=10, vbr=0) for _ in range(S)]
    sigs = [signals.music(10, channels=channels, seed=300 + s) for s in range(S)]
    sizes = set()
    for i in range(10 * 960 // frame):
        pcm = np.stack([np.ascontiguousarray(sigs[s][i * frame:(i + 1) * frame]).reshape(-1) for s in range(S)])
        pk, lens, rng = b.encode(pcm, frame)
        for s in range(S):
            a = chk[s].encode(np.ascontiguousarray(sigs[s][i * frame:(i + 1) * frame]), frame)
            assert (a[0], a[1], a[2]) == (pk[s], int(lens[s]), int(rng[s])), (i, s, a[1], int(lens[s]))
            sizes.add(int(lens[s]))
    assert len(sizes) == 1
    b.close()

def test_gpu_full_size_properties():
    """BASELINE size (65,536 streams): size-independent properties — identical inputs give identical packets on every
    wavefront; a sampled subset matches the oracle; every packet decodes with matching final range is checked on the subset
    through the reference decoder when available."""
    oa = _oa()
    from test_oracle_encoder import OracleEnc
    S = 65536
    b = oa.EncoderBatch(S, channels=2); b.ctl(oa.OPUS_SET_BITRATE_REQUEST, 128000); b.ctl(oa.OPUS_SET_COMPLEXITY_REQUEST, 10)
    base = [signals.music(4, seed=k) for k in range(8)]
    o = [OracleEnc(2, bitrate=128000, complexity=10) for _ in range(8)]
    for i in range(3):
        fr = np.stack([base[k][i * 960:(i + 1) * 960].reshape(-1) for k in range(8)])
        pcm = np.tile(fr, (S // 8, 1))
        pk, lens, rng = b.encode(pcm, 960)
        for k in range(8):
            a = o[k].encode(np.ascontiguousarray(base[k][i * 960:(i + 1) * 960]), 960)
            idx = np.arange(k, S, 8)
            assert np.all(lens[idx] == a[1]) and np.all(rng[idx] == a[2])
            for s in (k, k + 8 * 1000, k + 8 * 8191): assert pk[s] == a[0]
    b.close()


@pytest.mark.parametrize("name,Fs,ch,app,ctl", [
    ("config 3", 16000, 1, 2048, ((11002, 1000), (4008, 1103), (4002, 24000), (4010, 10))),
    ("config 4", 48000, 2, 2049, ((11002, 1001), (4008, 1105), (4002, 128000), (4010, 10)))])
def test_gpu_full_size_properties_silk_and_hybrid(name, Fs, ch, app, ctl):
    """the SILK-capable kernel at BASELINE width: 65,536 streams of config 3 / config 4, 32 distinct signals tiled over them, three consecutive frames with the state
    carried on the device; every replica must agree with the compiled reference encoder in length and final range, sampled replicas byte for byte"""
    import capi
    from test_kernel_emu_silkdec import speechy
    oa = _oa()
    S, U, n = 65536, 32, Fs // 50
    b = oa.EncoderBatch(S, channels=ch, application=app, Fs=Fs)
    for req, v in ctl: b.ctl(req, v)
    step = 48000 // Fs
    base = [np.ascontiguousarray(speechy(5, ch, 400 + k, 960)[::step]) for k in range(U)]
    req_name = {11002: "force_mode", 4008: "bandwidth", 4002: "bitrate", 4010: "complexity"}
    refs = [capi.Enc("ref", Fs, ch, app, **{req_name[r]: v for r, v in ctl}) for _ in range(U)]
    for i in range(3):
        fr = np.stack([base[k][i * n:(i + 1) * n].reshape(-1) for k in range(U)])
        pk, lens, rng = b.encode(np.tile(fr, (S // U, 1)), n)
        for k in range(U):
            a = refs[k].encode(np.ascontiguousarray(base[k][i * n:(i + 1) * n]), n)
            idx = np.arange(k, S, U)
            assert np.all(lens[idx] == a[1]) and np.all(rng[idx] == a[2]), (name, i, k)
            for s in (k, k + U * 1000, k + U * 2047): assert pk[s] == a[0], (name, i, k, s)
    b.close()


def test_gpu_parity_soak_with_midstream_ctls():
    """the parity gate of SURVEY 8d inside the suite: configs 2, 3 and 4, 256 streams x 1000 consecutive frames each, 64 distinct base signals (the rest are
    shifted / scaled copies), bitrate / complexity / VBR / FEC / DTX / bandwidth / channel changes applied mid-stream to the batch and to every reference
    encoder at the same frame (tools/parity_soak.py SCHEDULE): every packet, length and final range equal.  Round 3: the encoder runs with its tonality / music analysis
    (the default) and the reference is the build with the float API (libopus_ref_fxa.so), the fixed-point library as it is built by default."""
    import subprocess, sys, json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tools/parity_soak.py"), "--streams", "256", "--frames", "1000", "--bases", "64", "--ctl-schedule", "--float-analysis"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1500)
    out = p.stdout.decode(errors="replace")
    assert p.returncode == 0, out[-3000:]
    rows = [json.loads(l) for l in out.splitlines() if l.startswith("{")]
    assert len(rows) == 3, out[-3000:]
    for r in rows: assert r["mismatches"] == 0 and r["stream_frames_checked"] == 256000, r


def test_gpu_parity_soak_without_the_float_api():
    """the same gate for config 2 -- the configuration whose packets the analysis changes -- with the private switch off, against the reference built with
    DISABLE_FLOAT_API: 256 streams x 400 frames with the control schedule's first two changes"""
    import subprocess, sys, json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tools/parity_soak.py"), "--streams", "256", "--frames", "400", "--bases", "64", "--ctl-schedule", "--configs", "2"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    out = p.stdout.decode(errors="replace")
    assert p.returncode == 0, out[-3000:]
    rows = [json.loads(l) for l in out.splitlines() if l.startswith("{")]
    assert len(rows) == 1 and rows[0]["mismatches"] == 0 and rows[0]["stream_frames_checked"] == 256 * 400, out[-3000:]
