"""The decoder's CELT-only fast kernel in front of the general kernel against the general kernel alone, on the CPU wave emulator: identical PCM, sample counts, final ranges and
stream records over sequences that move streams between the two kernels (tools/dec_fast_check.py).  The emulator watches the fast kernel's dynamic LDS: it must not touch the A arena."""
import os, sys, pytest
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from reflib import ref_fx
pytestmark = pytest.mark.skipif(ref_fx() is None, reason="oracle/_ref not built")

def test_emu_fast_decoder_equals_general_decoder(tmp_path):
    import dec_fast_check
    assert not dec_fast_check.compare("emu", tmpdir=str(tmp_path), verbose=False)
