"""MI355X: the decoder's CELT-only fast kernel in front of the general kernel against the general kernel alone (tools/dec_fast_check.py)"""
import os, sys, pytest
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
pytestmark = pytest.mark.gpu

def test_gpu_fast_decoder_equals_general_decoder(tmp_path):
    import dec_fast_check
    assert not dec_fast_check.compare("gpu", tmpdir=str(tmp_path), verbose=False)

def test_gpu_fast_decoder_with_the_pvq_stage_forced_equals_general_decoder(tmp_path, monkeypatch):
    """the same with OPUS_AMD_DEC_PVQ4=1: the five-stream test batches through oa_celt_dpvq_kernel / oa_celt_dback_kernel (wide calls take them by default)"""
    monkeypatch.setenv("OPUS_AMD_DEC_PVQ4", "1")
    import importlib, dec_fast_check
    importlib.reload(dec_fast_check)
    assert not dec_fast_check.compare("gpu", tmpdir=str(tmp_path), verbose=False)

def test_gpu_pvq_stage_switch_per_batch():
    import test_hostemu_dec_fast as t
    t.pvq_stage_switch(os.path.join(ROOT, "opus_amd/libopus_amd.so"))

def test_gpu_pvq_stage_vs_reference():
    from reflib import ref_fx
    if ref_fx() is None: pytest.skip("compiled reference did not travel")
    import test_hostemu_dec_fast as t
    t.pvq_stage_vs_reference(os.path.join(ROOT, "opus_amd/libopus_amd.so"))
