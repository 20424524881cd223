"""bench.py's default corpus for the 48 kHz configurations: its numpy restatement of the reference's generate_music() (tests/test_opus_encode.c:57-85, driven by fast_rand(),
tests/test_opus_common.h:56-62) against the C function itself, compiled from the reference's source lines where they lie (skipped when /root/reference is absent)."""
import os, subprocess, sys, tempfile
import numpy as np, pytest
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference/tests/test_opus_encode.c"

def _c_generate_music(seed, n):
    src = open(REF).read().splitlines()
    a = next(i for i, l in enumerate(src) if l.startswith("void generate_music("))
    b = next(i for i in range(a, len(src)) if src[i].startswith("}"))
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "gm.c")
        open(c, "w").write("#include <stdio.h>\n#include <stdlib.h>\ntypedef int opus_int32; typedef unsigned opus_uint32;\n"
                           "static opus_uint32 Rz, Rw; static opus_uint32 fast_rand(void){ Rz=36969*(Rz&65535)+(Rz>>16); Rw=18000*(Rw&65535)+(Rw>>16); return (Rz<<16)+Rw; }\n"
                           + "\n".join(src[a:b + 1]) +
                           "\nint main(int c,char**v){ int n=atoi(v[2]); Rz=Rw=atoi(v[1]); short*b=malloc(4*n); generate_music(b,n); fwrite(b,4,n,stdout); return 0; }\n")
        exe = os.path.join(d, "gm")
        subprocess.check_call(["gcc", "-O1", "-o", exe, c])
        return np.frombuffer(subprocess.check_output([exe, str(seed), str(n)]), np.int16).reshape(-1, 2)

@pytest.mark.skipif(not os.path.exists(REF), reason="needs the reference tree")
def test_reference_music_matches_the_c_function():
    import bench
    n = 30000
    x = bench.reference_music(n - 2880, [42, 43, 1234567])
    for k, seed in enumerate((42, 43, 1234567)):
        r = _c_generate_music(seed, n)
        assert not r[:2880].any()                                   # the 60 ms of silence the bench skips
        assert np.array_equal(r[2880:], x[k])

@pytest.mark.skipif(not os.path.exists(REF), reason="needs the reference tree")
def test_reference_music_entered_late_is_the_same_piece():
    """a tune entered at sample `start` plays the reference's melody from there: once the two one-pole filters have settled it differs from the C output only by the dither"""
    import bench
    r = _c_generate_music(45, 60000).astype(np.int64)
    for start in (2887, 30001, 40000):
        y = bench.reference_music(3000, [45], starts=[start])[0].astype(np.int64)
        d = np.abs(y[500:] - r[start + 500:start + 3000])
        assert d.max() < 1200 and d.mean() < 200, (start, d.max(), d.mean())
    pool = bench.synth(bench.CONFIGS[2], 2, 8, 0, corpus="reference")
    assert pool.shape == (8, 4 * 960 * 2) and pool.dtype == np.int16 and len({p.tobytes() for p in pool}) == 8
