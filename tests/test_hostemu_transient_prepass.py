"""CPU: the transient analysis' recursions as a lane pre-pass ahead of the CELT-only encode kernel (celt_enc_front.h: ct_transient_tile) on the wave emulator against the compiled
reference.  (OPUS_AMD_SET_TRANSIENT_PREPASS(1) on the batch: a per-batch switch since round 6.)"""
import os, subprocess, sys, pytest
from reflib import ref_fx
pytestmark = pytest.mark.skipif(ref_fx() is None, reason="oracle/_ref not built")

def test_transient_prepass_matches_the_reference():
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "transient_prepass_check.py"), "emu"], env=dict(os.environ, OPUS_AMD_FLOAT_ANALYSIS="0"), capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
