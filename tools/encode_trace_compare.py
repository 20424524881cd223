#!/usr/bin/env python3
"""tools/encode_trace_compare.py <a.log> <b.log> — compare two logs of tools/encode_trace_shim.c call by call (the shorter one may still be running: its length is
compared as a prefix) and print the first differing calls."""
import sys
import gzip
def rd(p): return [l for l in (gzip.open(p, 'rt') if p.endswith('.gz') else open(p)).read().splitlines() if not l.startswith('#')]      # context lines (creation, controls) are not calls
a = rd(sys.argv[1]); b = rd(sys.argv[2])
n = min(len(a), len(b)); bad = [i for i in range(n) if a[i] != b[i]]
print("calls: %d / %d, compared %d, differing %d" % (len(a), len(b), n, len(bad)))
for i in bad[:10]: print("  ", a[i], "|", b[i])
sys.exit(1 if bad or len(a) != len(b) else 0)
