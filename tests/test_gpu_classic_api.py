"""MI355X: the classic-API parity tests of tests/test_hostemu_encoder_modes.py and tests/test_hostemu_decoder_rates.py (reference vs product, packet bytes / PCM and
final range call by call) with the product library opus_amd/libopus_amd.so in place of the emulated C ABI: mode switches with redundancy and prefills, multi-frame
packets, API rates below 48 kHz in both directions, CBR padding, settings fuzz."""
import pytest
import test_hostemu_encoder_modes as M, test_hostemu_decoder_rates as D, test_hostemu_api_limits as A
pytestmark = pytest.mark.gpu

@pytest.fixture(autouse=True)
def _product_library(monkeypatch):
    monkeypatch.setattr(M, "WHICH", "gpu"); monkeypatch.setattr(D, "WHICH", "gpu"); monkeypatch.setattr(A, "WHICH", "gpu")

from test_hostemu_encoder_modes import (test_silk_celt_switches_mono, test_hybrid_celt_switches_stereo, test_switches_10ms_and_short_frames, test_auto_mode_rate_sweep,
    test_silk_bandwidth_switch, test_long_frames_celt_and_hybrid, test_long_frames_silk, test_celt_below_48k, test_cbr_padding_and_tiny_buffers, test_settings_fuzz)
from test_hostemu_decoder_rates import test_celt_rates, test_silk_rates, test_hybrid_and_switches
from test_hostemu_api_limits import (test_hard_cbr_above_510_kbps, test_bitrate_max_pads_a_long_call_to_the_whole_buffer, test_packet_pad_to_tens_of_kilobytes,
    test_ms_encoder_batch_follows_the_encoder_in_use, test_ms_hard_cbr_above_the_frame_cap, test_ms_decoder_gain_and_fec_frame_size_check,
    test_ms_decode_sub_packet_longer_than_six_frames)
