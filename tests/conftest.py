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


# The two long reference programs of tests/test_zz_reference_programs.py (test_opus_decode, and test_opus_encode under the trace shim: ~7 and ~9 minutes, each bound by the
# latency of one-wave launches, not by the GPU) are started when a "-m gpu" session starts and run beside the rest of the suite; the test that owns them collects them at the
# end.  Without this the GPU suite is ~6 minutes longer.  (Anything else -- a single test selected by hand, no device, binaries missing -- and the test runs them itself.)
BACKGROUND = {}
def pytest_sessionstart(session):
    try:
        if session.config.getoption("markexpr", "") != "gpu" or session.config.getoption("keyword", ""): return
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        exe = {n: os.path.join(root, "oracle/_ref/reftests/gpu", n) for n in ("test_opus_decode", "test_opus_encode")}
        shim = os.path.join(root, "oracle/_ref/enc_trace_shim.so")
        if not all(os.path.exists(x) for x in list(exe.values()) + [shim, os.path.join(root, "opus_amd/libopus_amd.so")]): return
        import subprocess, tempfile
        d = tempfile.mkdtemp(prefix="oa_refprog_")
        env = dict(os.environ, OPUS_AMD_FLOAT_ANALYSIS="1", SEED="20260922")
        BACKGROUND["dir"] = d
        BACKGROUND["test_opus_decode"] = subprocess.Popen([exe["test_opus_decode"]], stdout=open(os.path.join(d, "dec.out"), "wb"), stderr=subprocess.STDOUT, env=env)
        BACKGROUND["test_opus_encode"] = subprocess.Popen([exe["test_opus_encode"]], stdout=open(os.path.join(d, "enc.out"), "wb"), stderr=subprocess.STDOUT,
                                                          env=dict(env, LD_PRELOAD=shim, OPUS_TRACE_FILE=os.path.join(d, "enc_trace.log")))
    except Exception:
        BACKGROUND.clear()

def pytest_sessionfinish(session, exitstatus):
    for k in ("test_opus_decode", "test_opus_encode"):
        p = BACKGROUND.get(k)
        if p is not None and p.poll() is None: p.kill()
