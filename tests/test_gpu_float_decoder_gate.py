"""MI355X: packets of the GPU encoder through the reference's float decoder (tests/float_gate_check.py)"""
import pytest
import float_gate_check as G
pytestmark = pytest.mark.gpu

@pytest.mark.parametrize("case", range(len(G.CASES)))
def test_gpu_packets_decode_with_float_reference(case): G.check("gpu", *G.CASES[case], frames=25)
