"""CPU: the device-resident multistream batch (opusgpu_ms_enc_batch_*) on the wave emulator against the compiled reference's opus_multistream_encode"""
import pytest
from reflib import ref_fx
import ms_batch_check
pytestmark = pytest.mark.skipif(ref_fx() is None, reason="oracle/_ref not built")

@pytest.mark.parametrize("case", range(len(ms_batch_check.CASES)))
def test_emu_ms_batch(case): ms_batch_check.check("emu", **ms_batch_check.CASES[case])
