"""MI355X: the decoder's CELT-only fast kernel in front of the general kernel against the general kernel alone (tools/dec_fast_check.py)"""
import os, sys, pytest
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
pytestmark = pytest.mark.gpu

def test_gpu_fast_decoder_equals_general_decoder(tmp_path):
    import dec_fast_check
    assert not dec_fast_check.compare("gpu", tmpdir=str(tmp_path), verbose=False)
