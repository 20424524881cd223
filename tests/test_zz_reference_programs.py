"""The reference's own UNMODIFIED C test programs (tests/test_opus_api.c, test_opus_padding.c, test_opus_projection.c, test_opus_decode.c, test_opus_encode.c + opus_encode_regressions.c,
src/opus_demo.c), compiled where they lie by tests/hostemu.py against this library's C ABI and run:
  * here (no GPU) against the emulated library tests/emu/libopus_amd_emu.so: the quick ones on every run; the long ones (tens of minutes on the CPU wave emulator)
    when OPUS_AMD_LONG_TESTS=1;
  * on the MI355X (-m gpu) against opus_amd/libopus_amd.so, binaries prebuilt under oracle/_ref/reftests/gpu/ by __graft_entry__.build().
Exit status 0 is the reference's own pass criterion (every failed check calls abort())."""
import os, subprocess, pytest
import hostemu
ROOT = hostemu.ROOT
LONG = os.environ.get("OPUS_AMD_LONG_TESTS") == "1"
# these programs run the library the way a user gets it: the tonality / music analysis on (opus_demo's default complexity is 10), whatever the rest of the session uses;
# the comparison side of the opus_demo tests is linked against the reference built with its float API (libopus_ref_fxa.so)
ENV = dict(os.environ, OPUS_AMD_FLOAT_ANALYSIS="1")

def _run(flavour, name, timeout, args=()):
    exe = os.path.join(ROOT, "oracle/_ref/reftests", flavour, name)
    if not os.path.exists(exe):
        if not os.path.isdir(hostemu.REF): pytest.skip("reference test binaries not built and /root/reference absent")
        hostemu.build_reftests(flavour)
    p = subprocess.run([exe] + list(args), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout, env=dict(ENV, SEED="20260922"))
    assert p.returncode == 0, p.stdout.decode(errors="replace")[-3000:]
    return p.stdout.decode(errors="replace")

@pytest.mark.skipif(not os.path.isdir(hostemu.REF) and not os.path.exists(os.path.join(ROOT, "oracle/_ref/reftests/emu/test_opus_api")), reason="no reference tree")
def test_emu_test_opus_api(): _run("emu", "test_opus_api", 1800)
@pytest.mark.skipif(not os.path.isdir(hostemu.REF) and not os.path.exists(os.path.join(ROOT, "oracle/_ref/reftests/emu/test_opus_api_fl")), reason="no reference tree")
def test_emu_test_opus_api_with_the_float_api(): _run("emu", "test_opus_api_fl", 1800)          # the same program compiled without -DDISABLE_FLOAT_API: its float-entry checks run
@pytest.mark.skipif(not os.path.isdir(hostemu.REF) and not os.path.exists(os.path.join(ROOT, "oracle/_ref/reftests/emu/test_opus_padding")), reason="no reference tree")
def test_emu_test_opus_padding(): _run("emu", "test_opus_padding", 600)
@pytest.mark.skipif(not os.path.isdir(hostemu.REF) and not os.path.exists(os.path.join(ROOT, "oracle/_ref/reftests/emu/test_opus_projection")), reason="no reference tree")
def test_emu_test_opus_projection(): assert "All projection tests passed" in _run("emu", "test_opus_projection", 600)
@pytest.mark.skipif(not LONG, reason="tens of minutes on the CPU wave emulator: OPUS_AMD_LONG_TESTS=1")
def test_emu_test_opus_encode(): _run("emu", "test_opus_encode", 4 * 3600)
@pytest.mark.skipif(not LONG, reason="tens of minutes on the CPU wave emulator: OPUS_AMD_LONG_TESTS=1")
def test_emu_test_opus_decode(): _run("emu", "test_opus_decode", 4 * 3600)

@pytest.mark.gpu
def test_gpu_test_opus_api(): _run("gpu", "test_opus_api", 900)
@pytest.mark.gpu
def test_gpu_test_opus_api_with_the_float_api(): _run("gpu", "test_opus_api_fl", 900)
@pytest.mark.gpu
def test_gpu_test_opus_padding(): _run("gpu", "test_opus_padding", 300)
@pytest.mark.gpu
def test_gpu_test_opus_projection(): assert "All projection tests passed" in _run("gpu", "test_opus_projection", 300)
def _traced(flavour, tmp_path, timeout):
    """test_opus_encode IN FULL (mode matrix, multistream, frame-size switching, the settings fuzz, the regression cases) under tools/encode_trace_shim.c: beyond the
    program's own pass criterion (it decodes what it encoded), every packet of the run must be the one the reference produced in the same run -- the program is
    deterministic for a seed, and the reference's run (linked to oracle/_ref/libopus_ref_fxa.so in the build container) is committed as one SHA-1 per 1,000 calls
    (tests/golden/enc_trace_<seed>.digest, tools/enc_trace_digest.py).  This comparison is what found the 16-bit logSum() of the surround masks and the DTX gating of the
    CELT-only applications in round 3; neither made the program itself fail."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import enc_trace_digest
    shim = os.path.join(ROOT, "oracle/_ref/enc_trace_shim.so")
    if not os.path.exists(shim): hostemu.build_trace_shim()
    log = os.path.join(str(tmp_path), "enc_trace.log")
    exe = os.path.join(ROOT, "oracle/_ref/reftests", flavour, "test_opus_encode")
    p = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout, env=dict(ENV, SEED="20260922", LD_PRELOAD=shim, OPUS_TRACE_FILE=log))
    assert p.returncode == 0, p.stdout.decode(errors="replace")[-3000:]
    bad = enc_trace_digest.check(log, os.path.join(ROOT, "tests/golden/enc_trace_20260922.digest"))
    assert bad is None, bad

@pytest.mark.skipif(not LONG, reason="hours on the CPU wave emulator: OPUS_AMD_LONG_TESTS=1")
def test_emu_test_opus_encode_every_packet_is_the_references(tmp_path): _traced("emu", tmp_path, 12 * 3600)

@pytest.mark.gpu
def test_gpu_test_opus_decode_and_encode(tmp_path):
    """test_opus_decode in full and test_opus_encode in full with every packet checked against the reference's (see _traced).  The two programs run side by side: each is
    bound by the latency of single-wave launches (~1-2 ms per call, tools/classic_latency.py), not by the GPU.  Measured on the MI355X: test_opus_decode 6 min 45 s
    (profiles/r02_b), the traced test_opus_encode with its fuzz section ~9 min next to another process (208,835 encode calls at this seed; profiles/r03_m)."""
    import threading, sys
    import conftest
    bg = conftest.BACKGROUND
    if bg.get("test_opus_decode") is not None and bg.get("test_opus_encode") is not None:           # started with the session (tests/conftest.py): collect
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import enc_trace_digest
        for name, out in (("test_opus_decode", "dec.out"), ("test_opus_encode", "enc.out")):
            rc = bg[name].wait(timeout=1500)
            assert rc == 0, (name, open(os.path.join(bg["dir"], out), "rb").read().decode(errors="replace")[-3000:])
        bad = enc_trace_digest.check(os.path.join(bg["dir"], "enc_trace.log"), os.path.join(ROOT, "tests/golden/enc_trace_20260922.digest"))
        assert bad is None, bad
        return
    res = {}
    def dec():
        try:
            p = subprocess.run([os.path.join(ROOT, "oracle/_ref/reftests/gpu/test_opus_decode")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1500, env=dict(ENV, SEED="20260922"))
            res["test_opus_decode"] = (p.returncode, p.stdout.decode(errors="replace")[-2000:])
        except Exception as ex: res["test_opus_decode"] = (-1, repr(ex))
    def enc():
        try: _traced("gpu", tmp_path, 1500); res["test_opus_encode"] = (0, "")
        except BaseException as ex: res["test_opus_encode"] = (-1, repr(ex)[-3000:])
    if not os.path.exists(os.path.join(ROOT, "oracle/_ref/reftests/gpu/test_opus_decode")):
        if not os.path.isdir(hostemu.REF): pytest.skip("reference test binaries not built and /root/reference absent")
        hostemu.build_reftests("gpu")
    ts = [threading.Thread(target=dec), threading.Thread(target=enc)]
    for t in ts: t.start()
    for t in ts: t.join()
    for name, (rc, out) in res.items(): assert rc == 0, (name, out)


def _opus_demo_roundtrip(flavour, tmp_path, args, Fs, ch, seconds=1.0):
    """the reference's opus_demo (src/opus_demo.c, unmodified) linked against this library and against the reference's own: same .bit file out of the encoder, same PCM
    out of the decoder (opus_demo drives the 24-bit entry points and checks the final range of every packet itself)"""
    import numpy as np
    from test_kernel_emu_silkdec import speechy
    for fl in (flavour, "ref"):
        exe = os.path.join(ROOT, "oracle/_ref/reftests", fl, "opus_demo")
        if not os.path.exists(exe):
            if not os.path.isdir(hostemu.REF): pytest.skip("opus_demo binaries not built and /root/reference absent")
            hostemu.build_reftests(fl)
    n = int(Fs * seconds)
    sig = np.ascontiguousarray(speechy(n * (48000 // Fs) // 960 + 2, ch, 99, 960)[::48000 // Fs][:n]).astype("<i2")
    pcm = tmp_path / "in.pcm"; sig.tofile(pcm)
    outs = {}
    for fl in (flavour, "ref"):
        exe = os.path.join(ROOT, "oracle/_ref/reftests", fl, "opus_demo")
        bit = tmp_path / (fl + ".bit"); dec = tmp_path / (fl + ".dec")
        p = subprocess.run([exe, "-e"] + list(args) + [str(pcm), str(bit)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600, env=ENV)
        assert p.returncode == 0, p.stdout.decode(errors="replace")[-2000:]
        p = subprocess.run([exe, "-d", str(Fs), str(ch), str(bit), str(dec)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600, env=ENV)
        assert p.returncode == 0, p.stdout.decode(errors="replace")[-2000:]
        outs[fl] = (bit.read_bytes(), dec.read_bytes())
    assert len(outs["ref"][0]) > 1000
    assert outs[flavour][0] == outs["ref"][0], "bitstreams differ"
    assert outs[flavour][1] == outs["ref"][1], "decoded PCM differs"

DEMO_CASES = [(["audio", "48000", "2", "64000"], 48000, 2), (["voip", "16000", "1", "20000", "-inbandfec", "-loss", "10"], 16000, 1),
              (["restricted-lowdelay", "48000", "2", "96000", "-framesize", "10", "-cbr"], 48000, 2), (["audio", "24000", "1", "32000", "-framesize", "40"], 24000, 1)]

@pytest.mark.skipif(not os.path.isdir(hostemu.REF) and not os.path.exists(os.path.join(ROOT, "oracle/_ref/reftests/ref/opus_demo")), reason="no reference tree")
@pytest.mark.parametrize("case", range(len(DEMO_CASES)))
def test_emu_opus_demo_roundtrip(case, tmp_path): _opus_demo_roundtrip("emu", tmp_path, *DEMO_CASES[case], seconds=0.6)

@pytest.mark.gpu
@pytest.mark.parametrize("case", range(len(DEMO_CASES)))
def test_gpu_opus_demo_roundtrip(case, tmp_path): _opus_demo_roundtrip("gpu", tmp_path, *DEMO_CASES[case], seconds=2.0)


# opus_demo as encoder + decoder in one process (its own final-range check on every packet), with its built-in mode / bandwidth / frame-size schedules, random frame sizes,
# random FEC, simulated loss, DTX, bitrate sweeps: the decoded PCM of the run linked against this library must equal the run linked against the reference's
DEMO_MODES = [(48000, 2, ["audio", "48000", "2", "96000", "-hybrid48k_test"]), (48000, 2, ["audio", "48000", "2", "64000", "-celt_test"]),
              (48000, 2, ["audio", "48000", "2", "128000", "-celt_hq_test"]), (24000, 1, ["audio", "24000", "1", "32000", "-hybrid24k_test"]),
              (16000, 1, ["voip", "16000", "1", "20000", "-silk16k_test"]), (12000, 1, ["voip", "12000", "1", "16000", "-silk12k_test"]),
              (8000, 1, ["voip", "8000", "1", "12000", "-silk8k_test"]), (48000, 2, ["audio", "48000", "2", "48000", "-random_framesize", "-random_fec", "-loss", "5"]),
              (48000, 1, ["voip", "48000", "1", "24000", "-dtx", "-cvbr", "-framesize", "60"]),
              (48000, 2, ["restricted-lowdelay", "48000", "2", "256000", "-framesize", "2.5", "-max_payload", "400"]),
              (48000, 2, ["audio", "48000", "2", "32000", "-bandwidth", "SWB", "-forcemono", "-framesize", "80"]),
              (48000, 2, ["audio", "48000", "2", "80000", "-sweep", "2000", "-sweep_max", "160000", "-framesize", "120"]),
              (16000, 2, ["audio", "16000", "2", "40000", "-inbandfec", "-loss", "15", "-complexity", "5"])]

def _opus_demo_codec(flavour, tmp_path, Fs, ch, args, seconds):
    import numpy as np
    from test_kernel_emu_silkdec import speechy
    for fl in (flavour, "ref"):
        if not os.path.exists(os.path.join(ROOT, "oracle/_ref/reftests", fl, "opus_demo")):
            if not os.path.isdir(hostemu.REF): pytest.skip("opus_demo binaries not built and /root/reference absent")
            hostemu.build_reftests(fl)
    n = int(Fs * seconds)
    sig = np.ascontiguousarray(speechy(n * (48000 // Fs) // 960 + 2, ch, len(args), 960)[::48000 // Fs][:n]).astype("<i2")
    pcm = tmp_path / "in.pcm"; sig.tofile(pcm)
    outs = []
    for fl in (flavour, "ref"):
        o = tmp_path / (fl + ".pcm")
        p = subprocess.run([os.path.join(ROOT, "oracle/_ref/reftests", fl, "opus_demo")] + list(args) + [str(pcm), str(o)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900, env=ENV)
        assert p.returncode == 0, (fl, p.stdout.decode(errors="replace")[-1500:])
        outs.append(o.read_bytes())
    assert len(outs[1]) > 1000 and outs[0] == outs[1], "decoded PCM differs"

@pytest.mark.skipif(not os.path.isdir(hostemu.REF) and not os.path.exists(os.path.join(ROOT, "oracle/_ref/reftests/ref/opus_demo")), reason="no reference tree")
@pytest.mark.parametrize("case", range(len(DEMO_MODES)))
def test_emu_opus_demo_schedules(case, tmp_path): _opus_demo_codec("emu", tmp_path, *DEMO_MODES[case], seconds=2.2)

@pytest.mark.gpu
@pytest.mark.parametrize("case", range(len(DEMO_MODES)))
def test_gpu_opus_demo_schedules(case, tmp_path): _opus_demo_codec("gpu", tmp_path, *DEMO_MODES[case], seconds=3.0)
