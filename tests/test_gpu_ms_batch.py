"""MI355X: the device-resident multistream batch through the C ABI against the compiled reference's opus_multistream_encode; plus the config-5 shape at width"""
import pytest
import ms_batch_check
pytestmark = pytest.mark.gpu

@pytest.mark.parametrize("case", range(len(ms_batch_check.CASES)))
def test_gpu_ms_batch(case): ms_batch_check.check("gpu", **ms_batch_check.CASES[case])

def test_gpu_ms_batch_config5_width():
    """255 mono AUDIO streams per encoder (BASELINE config 5), 3 encoders, 64 kb/s per stream"""
    ms_batch_check.check("gpu", B=3, channels=255, streams=255, coupled=0, mapping=list(range(255)), application=2049, bitrate=255 * 64000, frames=2)

@pytest.mark.parametrize("case", range(len(ms_batch_check.DEC_CASES)))
def test_gpu_ms_decode_batch(case): ms_batch_check.check_ms_decode("gpu", **dict(ms_batch_check.DEC_CASES[case], frames=8))

def test_gpu_ms_decode_batch_config5_width():
    """255 mono streams per decoder, 3 decoders: the serial Appendix-B walk of 255 sub-packets per packet on lane 0, 765 elementary decodes per step"""
    ms_batch_check.check_ms_decode("gpu", B=3, channels=255, streams=255, coupled=0, mapping=list(range(255)), application=2049, bitrate=255 * 64000, frames=3)

@pytest.mark.parametrize("channels,analysis", [(4, False), (9, False), (6, True), (16, True), (38, False)])
def test_gpu_projection_batches(channels, analysis):
    """projection encoder batch (device mixing) and decoder batch (device demixing), orders 1-5 with and without the non-diegetic pair, against opus_projection_encode / _decode"""
    ms_batch_check.check_projection("gpu", B=3, channels=channels, bitrate=channels * 48000, complexity=10 if analysis else 5, analysis=analysis, frames=6)

@pytest.mark.parametrize("channels", [3, 4, 5, 6, 7, 8])
def test_gpu_surround_batch(channels): ms_batch_check.check_surround("gpu", B=3, channels=channels, bitrate=channels * 56000, frames=8)

def test_gpu_ms_decode_batch_turns_an_over_long_elementary_packet_away_whole(): ms_batch_check.check_ms_decode_slot_limit("gpu")

@pytest.mark.parametrize("case", range(len(ms_batch_check.TIGHT_CASES)))
def test_gpu_ms_batch_chained_byte_budgets(case): ms_batch_check.check("gpu", **ms_batch_check.TIGHT_CASES[case])

@pytest.mark.parametrize("case", range(len(ms_batch_check.CBR_CASES)))
def test_gpu_ms_batch_hard_cbr(case): ms_batch_check.check("gpu", **ms_batch_check.CBR_CASES[case])
