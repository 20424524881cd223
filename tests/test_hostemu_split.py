"""The SILK-capable encoder's kernel pipeline (front / quantiser / back: opus_amd/csrc/opus_sh_split.h) against its one-kernel path on the CPU wave emulator: identical packets,
final ranges AND stream records byte for byte over the case matrix of tools/split_check.py (configs 3 / 4, automatic modes, CBR, tight CVBR, 10 ms NB / MB with per-stream
settings, stereo SILK, calls the front kernel hands back), in both quantiser forms.  The emulator also watches every launch's dynamic LDS (hip_stub.h) and runs the matrix a
second time with positive garbage in "device" memory (OA_EMU_FILL=0x5A: an index read from an unloaded word faults here as it would on the GPU)."""
import os, sys, pytest
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from reflib import ref_fx
pytestmark = pytest.mark.skipif(ref_fx() is None, reason="oracle/_ref not built")

LONG = os.environ.get("OPUS_AMD_LONG_TESTS") == "1"        # the default CPU suite: one garbage pattern, the cases that reach every kind of call; OPUS_AMD_LONG_TESTS=1: both patterns, the whole matrix, the wider fuzz
QUICK = "config3,config4,audio_auto,audio_celt,switching,10ms_nb_mb"

@pytest.mark.parametrize("fill", ["0xA5", "0x5A"] if LONG else ["0x5A"])
def test_emu_split_path_equals_one_kernel_path(fill, monkeypatch, tmp_path):
    import split_check
    monkeypatch.setenv("OA_EMU_FILL", fill)
    if not LONG: monkeypatch.setenv("SPLIT_CHECK_CASES", QUICK)
    bad, stats = split_check.compare("emu", tmpdir=str(tmp_path), verbose=False)
    assert not bad, bad
    assert stats["config3"] == (444, 0) and stats["config4"] == (152, 0) and stats["audio_auto"] == (84, 0)      # (kept, handed back): the forced modes keep every call; so do the automatic ones since in-band FEC goes through the pipeline (round 6)
    assert stats["10ms_nb_mb"][1] > 0                                                                           # complexity 0 is still handed back
    assert stats["audio_celt"] == (84, 0) and stats["switching"][0] > 90                                        # CELT-only calls stay in the pipeline, and so do most calls around the mode switches

def test_emu_settings_fuzzers_through_the_pipeline():
    """the settings fuzzers (encoder, sparse settings with resets, multistream layouts, batch ABI: tests/test_hostemu_fuzz.py) and the unforced-mode / long-frame cases of the mode and analysis suites once more with
    OPUS_AMD_SH_SPLIT=1, which sends every 10 / 20 ms call of the SILK-capable applications through the front / quantiser / back kernels whatever the width of the launch: the
    pipeline -- CELT-only frames kept by the front kernel, transitions, redundancy, declined calls -- against the compiled reference, not only against the one-kernel path"""
    import subprocess
    env = dict(os.environ, OPUS_AMD_SH_SPLIT="1")
    p = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", "-m", "not gpu", "-k", "settings_fuzz or unforced_mode or long_frames" if LONG else "test_settings_fuzz or unforced_mode",
                        os.path.join(ROOT, "tests/test_hostemu_fuzz.py"), os.path.join(ROOT, "tests/test_hostemu_encoder_modes.py"), os.path.join(ROOT, "tests/test_hostemu_analysis.py")],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env, cwd=ROOT, timeout=3000)
    assert p.returncode == 0, p.stdout.decode(errors="replace")[-3000:]
