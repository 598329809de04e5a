"""Instruction mix of the hottest loop of a kernel: python tools/isa_loop.py file.s <substring of mangled name>
The loop is taken as the span between the backward branch with the largest distance and its target label."""
import re, sys, collections
lines = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2]
start = next(i for i, l in enumerate(lines) if re.match(r"_Z\S*:", l) and pat in l)
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
labels = {m.group(1): i for i in range(start, end) if (m := re.match(r"(\.LBB\d+_\d+):", lines[i]))}
best = None
for i in range(start, end):
    m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", lines[i]) or re.match(r"\s+s_branch\s+(\.LBB\d+_\d+)", lines[i])
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        span = (labels[m.group(1)], i)
        n_mfma = sum("v_mfma" in l for l in lines[span[0]:span[1]])
        if best is None or n_mfma > best[0]:
            best = (n_mfma, span)
n, (a, b) = best
c = collections.Counter()
for l in lines[a:b]:
    m = re.match(r"\s+([a-z][a-z0-9_]+)\s", l)
    if m: c[m.group(1)] += 1
valu = sum(v for k, v in c.items() if k.startswith("v_") and not k.startswith("v_mfma"))
print(f"loop lines {a - start}..{b - start}: mfma {n} valu {valu} salu {sum(v for k, v in c.items() if k.startswith('s_'))} "
      f"ds {sum(v for k, v in c.items() if k.startswith('ds_'))} vmem {sum(v for k, v in c.items() if k.startswith(('global_', 'buffer_', 'scratch_')))}")
print(", ".join(f"{k}:{v}" for k, v in c.most_common(40)))
