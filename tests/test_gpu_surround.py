"""MI355X: tests/test_hostemu_surround.py with the product library opus_amd/libopus_amd.so in place of the emulated C ABI -- the surround masking analysis against the
reference's surround_analysis, and whole surround encoders (3 to 8 channels, every application, API rates 8-48 kHz, with and without the float-API analysis) against
the compiled reference, packet by packet; and tests/test_hostemu_projection.py: projection (mapping family 3) encoders and decoders of every ambisonics order."""
import pytest
import test_hostemu_surround as S, test_hostemu_projection as P
pytestmark = pytest.mark.gpu

@pytest.fixture(autouse=True)
def _product_library(monkeypatch):
    monkeypatch.setattr(S, "WHICH", "gpu"); monkeypatch.setattr(P, "WHICH", "gpu")

from test_hostemu_surround import (test_surround_analysis_against_the_reference_function, test_surround_encoders_every_layout, test_surround_rates_applications_and_settings,
    test_surround_with_the_float_api_analysis)
from test_hostemu_projection import test_projection_encoder_and_decoder, test_projection_rejects_what_the_reference_rejects, test_projection_float_entry_points
