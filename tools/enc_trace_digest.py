#!/usr/bin/env python3
"""tools/enc_trace_digest.py — condense a log of tools/encode_trace_shim.c (one line per opus_encode / opus_multistream_encode call of the reference's test_opus_encode:
frame size, byte budget, return value, TOC, packet hash) into one SHA-1 per 1,000 calls, so that the reference's whole run fits in a small committed file
(tests/golden/enc_trace_<seed>.digest) that the GPU box can check this library's run against without the reference being there.
   python tools/enc_trace_digest.py make <seed>          # build container: runs the reference-linked program (oracle/_ref) under the shim, writes the golden file
   python tools/enc_trace_digest.py check <log> <golden> # anywhere: exit 0 if the log has exactly the golden digests; prints the first differing block otherwise"""
import hashlib, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BLOCK = 1000

def digests(path):
    out = []; h = hashlib.sha1(); n = 0
    with open(path) as f:
        for line in f:
            if line.startswith("#"): continue                      # context lines (creation, controls)
            h.update(line.encode()); n += 1
            if n % BLOCK == 0: out.append("%d %s" % (n, h.hexdigest())); h = hashlib.sha1()
    if n % BLOCK: out.append("%d %s" % (n, h.hexdigest()))
    return out

def check(log, golden):
    got = digests(log); want = open(golden).read().split("\n")[1:]
    want = [w for w in want if w]
    for i, (g, w) in enumerate(zip(got, want)):
        if g != w: return "calls %d..%d differ from the reference's (or the runs have different lengths: %s vs %s)" % (i * BLOCK, (i + 1) * BLOCK - 1, g.split()[0], w.split()[0])
    if len(got) != len(want): return "run length differs: %d blocks against the reference's %d" % (len(got), len(want))
    return None

if __name__ == "__main__":
    if sys.argv[1] == "make":
        seed = sys.argv[2]; sys.path.insert(0, os.path.join(ROOT, "tests"))
        from hostemu import REFTEST_FLAGS, REF
        tmp = tempfile.mkdtemp()
        exe = os.path.join(tmp, "test_opus_encode_ref"); shim = os.path.join(tmp, "shim.so"); log = os.path.join(tmp, "ref.log")
        srcs = [os.path.join(REF, x) for x in ("tests/test_opus_encode.c", "tests/opus_encode_regressions.c")]
        subprocess.check_call(["gcc"] + REFTEST_FLAGS + srcs + ["-o", exe, "-L" + os.path.join(ROOT, "oracle/_ref"), "-l:libopus_ref_fxa.so", "-Wl,-rpath," + os.path.join(ROOT, "oracle/_ref"), "-lm"])
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", os.path.join(ROOT, "tools/encode_trace_shim.c"), "-o", shim, "-ldl"])
        subprocess.check_call([exe], env=dict(os.environ, SEED=seed, LD_PRELOAD=shim, OPUS_TRACE_FILE=log), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        d = digests(log)
        out = os.path.join(ROOT, "tests/golden/enc_trace_%s.digest" % seed)
        open(out, "w").write("# SHA-1 per %d encode calls of the reference's tests/test_opus_encode.c (SEED=%s, fuzz on) linked to oracle/_ref/libopus_ref_fxa.so: tools/enc_trace_digest.py make %s\n" % (BLOCK, seed, seed) + "\n".join(d) + "\n")
        print(out, len(d), "blocks")
    else:
        r = check(sys.argv[2], sys.argv[3]); print(r or "identical"); sys.exit(1 if r else 0)
