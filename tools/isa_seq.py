#!/usr/bin/env python3
"""usage: tools/isa_seq.py <file.s> <kernel-name-substring> — counts and run-length instruction sequence of one kernel's ISA."""
import re, sys
s = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
start = next(i for i, l in enumerate(s) if key in l and not l.startswith("\t") and l.split(";")[0].strip().endswith(":"))
end = next(i for i in range(start, len(s)) if "s_endpgm" in s[i])
body = s[start:end]
pats = ["v_mfma", "v_accvgpr_read", "v_accvgpr_write", "ds_read_b128", "ds_read", "ds_write", "global_load", "global_store", "s_barrier",
        "v_cndmask", "scratch_", "s_nop"]
print({p: sum(p in l for l in body) for p in pats}, "lines", len(body))
if len(sys.argv) > 3:
    seq = []
    for i, l in enumerate(body):
        t = l.strip().split(" ")[0].split("\t")[0]
        m = re.match(r"(global_load|global_store|s_waitcnt|s_barrier|v_mfma|ds_write|ds_read|s_cbranch|\.LBB)\S*", l.strip())
        if not m: continue
        k = m.group(1) if m.group(1) != "s_waitcnt" else l.strip()
        if m.group(1) in ("s_cbranch", ".LBB"): k = l.strip().split(";")[0]
        if seq and seq[-1][0] == k: seq[-1][1] += 1
        else: seq.append([k, 1, i])
    for k, n, i in seq[: int(sys.argv[3])]: print(i, k, "x%d" % n)
