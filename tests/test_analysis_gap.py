"""Where the build this port is bit-exact to (FIXED_POINT + DISABLE_FLOAT_API) and the fixed-point build users get by default (float API on: src/analysis.c + mlp.c
steer the encoder) agree and where they do not -- both sides are the compiled reference (oracle/_ref/libopus_ref_fx.so vs libopus_ref_fxa.so), tools/analysis_gap.py.
The forced SILK-only (BASELINE config 3) and forced hybrid (config 4) encoders and an unforced VOIP encoder produce IDENTICAL packets with and without the analysis:
for those the parity this repo proves against the no-float-API library is parity with the deployed fixed-point library as well.  CELT-coded frames at complexity >= 7
(config 2, unforced AUDIO) differ: that is the row DESIGN.md section 8 lists next."""
import os, sys, pytest
from reflib import ref_fx, ROOT
sys.path.insert(0, os.path.join(ROOT, "tools"))
pytestmark = pytest.mark.skipif(ref_fx() is None or not os.path.exists(os.path.join(ROOT, "oracle/_ref/libopus_ref_fxa.so")), reason="oracle/_ref not built")

def test_which_configurations_the_analysis_changes():
    import analysis_gap as G
    r = {c[0].split(":")[0].split(",")[0]: G.run(*c, frames=100) for c in G.CASES}
    for k in ("config 3", "config 4", "VOIP 16 kHz mono 20 kb/s"): assert r[k]["identical"] == r[k]["frames"], r[k]
    assert r["config 2"]["identical"] < r["config 2"]["frames"] and r["config 2"]["toc_differs"] == 0, r["config 2"]      # same decisions, different allocation tuning
