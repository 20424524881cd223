"""MI355X: the device-resident multistream batch through the C ABI against the compiled reference's opus_multistream_encode; plus the config-5 shape at width"""
import pytest
import ms_batch_check
pytestmark = pytest.mark.gpu

@pytest.mark.parametrize("case", range(len(ms_batch_check.CASES)))
def test_gpu_ms_batch(case): ms_batch_check.check("gpu", **ms_batch_check.CASES[case])

def test_gpu_ms_batch_config5_width():
    """255 mono AUDIO streams per encoder (BASELINE config 5), 3 encoders, 64 kb/s per stream"""
    ms_batch_check.check("gpu", B=3, channels=255, streams=255, coupled=0, mapping=list(range(255)), application=2049, bitrate=255 * 64000, frames=2)
