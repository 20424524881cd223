"""The reference's own UNMODIFIED C test programs (tests/test_opus_api.c, test_opus_padding.c, test_opus_decode.c, test_opus_encode.c + opus_encode_regressions.c,
src/opus_demo.c), compiled where they lie by tests/hostemu.py against this library's C ABI and run:
  * here (no GPU) against the emulated library tests/emu/libopus_amd_emu.so: the quick ones on every run; the long ones (tens of minutes on the CPU wave emulator)
    when OPUS_AMD_LONG_TESTS=1;
  * on the MI355X (-m gpu) against opus_amd/libopus_amd.so, binaries prebuilt under oracle/_ref/reftests/gpu/ by __graft_entry__.build().
Exit status 0 is the reference's own pass criterion (every failed check calls abort())."""
import os, subprocess, pytest
import hostemu
ROOT = hostemu.ROOT
LONG = os.environ.get("OPUS_AMD_LONG_TESTS") == "1"

def _run(flavour, name, timeout, args=()):
    exe = os.path.join(ROOT, "oracle/_ref/reftests", flavour, name)
    if not os.path.exists(exe):
        if not os.path.isdir(hostemu.REF): pytest.skip("reference test binaries not built and /root/reference absent")
        hostemu.build_reftests(flavour)
    p = subprocess.run([exe] + list(args), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout, env=dict(os.environ, SEED="20260922"))
    assert p.returncode == 0, p.stdout.decode(errors="replace")[-3000:]
    return p.stdout.decode(errors="replace")

@pytest.mark.skipif(not os.path.isdir(hostemu.REF) and not os.path.exists(os.path.join(ROOT, "oracle/_ref/reftests/emu/test_opus_api")), reason="no reference tree")
def test_emu_test_opus_api(): _run("emu", "test_opus_api", 1800)
@pytest.mark.skipif(not os.path.isdir(hostemu.REF) and not os.path.exists(os.path.join(ROOT, "oracle/_ref/reftests/emu/test_opus_padding")), reason="no reference tree")
def test_emu_test_opus_padding(): _run("emu", "test_opus_padding", 600)
@pytest.mark.skipif(not LONG, reason="tens of minutes on the CPU wave emulator: OPUS_AMD_LONG_TESTS=1")
def test_emu_test_opus_encode(): _run("emu", "test_opus_encode", 4 * 3600)
@pytest.mark.skipif(not LONG, reason="tens of minutes on the CPU wave emulator: OPUS_AMD_LONG_TESTS=1")
def test_emu_test_opus_decode(): _run("emu", "test_opus_decode", 4 * 3600)

@pytest.mark.gpu
def test_gpu_test_opus_api(): _run("gpu", "test_opus_api", 900)
@pytest.mark.gpu
def test_gpu_test_opus_padding(): _run("gpu", "test_opus_padding", 300)
@pytest.mark.gpu
def test_gpu_test_opus_decode(): _run("gpu", "test_opus_decode", 1500)
@pytest.mark.gpu
def test_gpu_test_opus_encode(): _run("gpu", "test_opus_encode", 2400)
