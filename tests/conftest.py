import os, sys, pytest
# The encoder's default is the reference's default: the tonality / music analysis of the float API runs at complexity 10 (checked against libopus_ref_fxa.so by the tests
# that switch it on explicitly).  The bulk of the suites check against the reference built with DISABLE_FLOAT_API and the plain-C restatement, which have no analysis:
# new encoders of this process start with the private switch off (opus_amd/csrc/opus_enc_host.h, OPUS_AMD_SET_FLOAT_ANALYSIS).
os.environ.setdefault("OPUS_AMD_FLOAT_ANALYSIS", "0")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")
    config.addinivalue_line("markers", "needs_ref: needs oracle/_ref/*.so (the compiled reference)")
