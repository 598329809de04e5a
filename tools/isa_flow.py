"""Compact schedule of one kernel in a hipcc -S listing: python tools/isa_flow.py file.s <substring of the mangled name>
M mfma, r/w LDS read/write, L/S/A global load/store/atomic, v/s vector/scalar ALU (runs counted), [..] s_waitcnt, {..} branches."""
import re, sys
lines = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2]
start = next(i for i, l in enumerate(lines) if re.match(r"_Z\S*:", l) and pat in l)
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
seq = []
for l in lines[start:end]:
    if re.match(r"\.LBB\S+:", l):
        seq.append("\n" + l.split(":")[0] + ("  (loop d%s)" % l.split("Depth=")[1][:1] if "Depth=" in l else "") + ": ")
        continue
    m = re.match(r"\s+([a-z_0-9]+)\s*(.*)", l)
    if not m:
        continue
    op, rest = m.group(1), m.group(2).strip()
    if op.startswith("v_mfma"): seq.append("M")
    elif op.startswith("ds_read") or op.startswith("ds_bperm") or op.startswith("ds_perm"): seq.append("r")
    elif op.startswith("ds_"): seq.append("w")
    elif op == "s_waitcnt": seq.append("[" + rest.replace("vmcnt", "vm").replace("lgkmcnt", "lgk").replace(" ", "") + "]")
    elif op.startswith("global_load") or op.startswith("buffer_load") or op.startswith("scratch_load"): seq.append("L" if not op.startswith("scratch") else "l")
    elif op.startswith("global_store") or op.startswith("buffer_store"): seq.append("S")
    elif op.startswith("scratch_store"): seq.append("z")
    elif op.startswith("global_atomic"): seq.append("A")
    elif op.startswith("s_cbranch") or op == "s_branch": seq.append("{" + rest.replace(".LBB", "") + "}")
    elif op.startswith("v_"): seq.append("v")
    elif op.startswith("s_"): seq.append("s")
    else: seq.append("?")
out = "".join(seq)
out = re.sub(r"v{3,}", lambda m: "v%d" % len(m.group(0)), out)
out = re.sub(r"s{3,}", lambda m: "s%d" % len(m.group(0)), out)
out = re.sub(r"M{3,}", lambda m: "M%d" % len(m.group(0)), out)
print(out)
