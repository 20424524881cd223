"""CPU: the CELT-only kernel pipeline -- the encode kernel cut before the PVQ, oa_celt_pvq_kernel with four streams per wave on 16-lane groups (opus_amd/csrc/celt_enc_pvq4.h),
oa_celt_back_kernel -- on the wave emulator against the compiled reference (tests/celt_pipe_check.py: packet bytes, lengths, final ranges), and the group collectives the
emulator gained for it."""
import os, subprocess, sys, pytest
from reflib import ref_fx
pytestmark = pytest.mark.skipif(ref_fx() is None, reason="oracle/_ref not built")
HERE = os.path.dirname(os.path.abspath(__file__))

@pytest.mark.parametrize("case", ["config2", "wide128", "corr64", "mono10ms", "low24", "c0", "cbr64", "tight", "fs24k", "short5ms"])
def test_celt_pipeline_matches_the_reference(case):
    r = subprocess.run([sys.executable, os.path.join(HERE, "celt_pipe_check.py"), "emu", case], env=dict(os.environ, OPUS_AMD_FLOAT_ANALYSIS="0"), capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and "1 cases" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]

@pytest.mark.parametrize("env", [dict(OA_EMU_REVERSE="1"), dict(OA_EMU_CHECK_UNI="1")], ids=["lanes-63-to-0", "uniformity-checks"])
def test_celt_pipeline_lane_order_and_uniformity(env):
    """the same with the emulator's fibers scheduled 63 .. 0 (a read of another lane's LDS write without a group sync in between shows) and with every wg_bcast index checked for
    uniformity inside its group"""
    r = subprocess.run([sys.executable, os.path.join(HERE, "celt_pipe_check.py"), "emu", "wide128"], env=dict(os.environ, OPUS_AMD_FLOAT_ANALYSIS="0", **env), capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and "1 cases" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
