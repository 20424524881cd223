"""MI355X: the SILK-capable encoder's kernel pipeline against its one-kernel path through the C ABI (tools/split_check.py): packets, final ranges and stream records identical"""
import os, sys, pytest
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
pytestmark = pytest.mark.gpu

def test_gpu_split_path_equals_one_kernel_path(tmp_path):
    import split_check
    bad, stats = split_check.compare("gpu", tmpdir=str(tmp_path), verbose=False)
    assert not bad, bad
    assert stats["config3"] == (444, 0) and stats["config4"] == (152, 0)
