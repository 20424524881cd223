"""MI355X: the transient analysis' recursions as a lane pre-pass ahead of the CELT-only encode kernel (celt_enc_front.h: ct_transient_tile) against the compiled reference --
tests/transient_prepass_check.py with the pre-pass switched on (OPUS_AMD_SET_TRANSIENT_PREPASS) for its narrow batches (a wide launch, >= 64 streams, takes it by itself: the bench's parity samples cover that)."""
import os, subprocess, sys, pytest
pytestmark = pytest.mark.gpu

def test_gpu_transient_prepass_matches_the_reference():
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "transient_prepass_check.py"), "gpu"], env=dict(os.environ, OPUS_AMD_FLOAT_ANALYSIS="0"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
