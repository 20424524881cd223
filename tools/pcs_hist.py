#!/usr/bin/env python3
"""tools/pcs_hist.py <pc_sampling csv> — histogram of rocprofv3 PC samples by (code object, offset): every distinct offset with its count (the file stays small whatever
the sample count; tools/pcs_funcs.py maps the offsets to functions and instructions of the library's code object)."""
import csv, sys, collections
h = collections.Counter(); cols = None; n = 0
with open(sys.argv[1]) as f:
    rd = csv.DictReader(f); cols = rd.fieldnames
    off = next((c for c in cols if "offset" in c.lower()), None); cid = next((c for c in cols if "code_object_id" in c.lower() or "Code_Object" in c), None)
    extra = [c for c in cols if any(k in c.lower() for k in ("stall", "inst_type", "issued", "reason", "exec_mask"))]
    for r in rd:
        n += 1
        h[(r.get(cid, "?") if cid else "?", r.get(off, "?") if off else "?") + tuple(r.get(c, "") for c in extra[:3])] += 1
print("# columns:", cols); print("# samples:", n, "key:", [cid, off] + extra[:3])
for k, v in sorted(h.items(), key=lambda kv: -kv[1]): print(v, *k)
