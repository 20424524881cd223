"""CPU: the device-resident multistream batch (opusgpu_ms_enc_batch_*) on the wave emulator against the compiled reference's opus_multistream_encode"""
import pytest
from reflib import ref_fx
import ms_batch_check
pytestmark = pytest.mark.skipif(ref_fx() is None, reason="oracle/_ref not built")

@pytest.mark.parametrize("case", range(len(ms_batch_check.CASES)))
def test_emu_ms_batch(case): ms_batch_check.check("emu", **ms_batch_check.CASES[case])

@pytest.mark.parametrize("case", range(len(ms_batch_check.DEC_CASES)))
def test_emu_ms_decode_batch(case): ms_batch_check.check_ms_decode("emu", **ms_batch_check.DEC_CASES[case])

@pytest.mark.parametrize("channels,analysis", [(4, False), (9, False), (6, True)])
def test_emu_projection_batches(channels, analysis):
    ms_batch_check.check_projection("emu", B=2, channels=channels, bitrate=channels * 48000, complexity=10 if analysis else 5, analysis=analysis)

@pytest.mark.parametrize("channels", [3, 6, 8])
def test_emu_surround_batch(channels): ms_batch_check.check_surround("emu", B=2, channels=channels, bitrate=channels * 56000)

def test_emu_ms_decode_batch_turns_an_over_long_elementary_packet_away_whole(): ms_batch_check.check_ms_decode_slot_limit("emu")

@pytest.mark.parametrize("case", range(len(ms_batch_check.TIGHT_CASES)))
def test_emu_ms_batch_chained_byte_budgets(case): ms_batch_check.check("emu", **ms_batch_check.TIGHT_CASES[case])

@pytest.mark.parametrize("case", range(len(ms_batch_check.CBR_CASES)))
def test_emu_ms_batch_hard_cbr(case): ms_batch_check.check("emu", **ms_batch_check.CBR_CASES[case])
