"""MI355X: the encoder with its tonality / music analysis (the default at complexity 10) against the reference built with the float API (oracle/_ref/libopus_ref_fxa.so):
tests/test_hostemu_analysis.py with the product library opus_amd/libopus_amd.so in place of the emulated C ABI -- here the analysis' IEEE arithmetic is the GPU's own
(no fused multiply-adds in it, correctly rounded division and square root, the device's double-precision log)."""
import pytest
import test_hostemu_analysis as A
pytestmark = pytest.mark.gpu

@pytest.fixture(autouse=True)
def _product_library(monkeypatch):
    monkeypatch.setattr(A, "WHICH", "gpu")

from test_hostemu_analysis import (test_config_2_with_analysis, test_lowdelay_rates_and_sizes, test_audio_unforced_mode_decisions, test_speech_music_speech,
    test_forced_modes_and_hybrid, test_dtx_with_activity_probability, test_cbr_and_constrained, test_controls_midstream, test_signal_type_steers_the_lowdelay_application,
    test_24_bit_and_float_entry_points_feed_the_analysis_unrounded_samples, test_multistream_with_analysis)
