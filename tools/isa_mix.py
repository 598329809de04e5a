"""Instruction mix of one kernel in a hipcc -S listing: python tools/isa_mix.py file.s <substring of mangled name>"""
import re, sys, collections
lines = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2]
start = next(i for i, l in enumerate(lines) if re.match(r"_Z\S*:", l) and pat in l)
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
c = collections.Counter()
for l in lines[start:end]:
    m = re.match(r"\s+([a-z][a-z0-9_]+)\s", l)
    if m: c[m.group(1)] += 1
tot = sum(c.values())
valu = sum(v for k, v in c.items() if k.startswith("v_") and not k.startswith("v_mfma"))
print(lines[start][:80]); print("total", tot, "mfma", sum(v for k, v in c.items() if k.startswith("v_mfma")), "valu", valu,
      "salu", sum(v for k, v in c.items() if k.startswith("s_")), "ds", sum(v for k, v in c.items() if k.startswith("ds_")))
print(", ".join(f"{k}:{v}" for k, v in c.most_common(45)))
