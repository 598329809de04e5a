#!/usr/bin/env python3
"""Summarise two rocprofv3 PMC passes of bench.py into profiles/<tag>_pmc_traffic.json.

On the GPU box (separate passes: FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2 — MI355X_MICROARCH.md):
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_f -o f --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fwd-eval
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_w -o w --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fwd-eval
then:  python profiles/collect_pmc.py gpurun_out/pmc_f/f_counter_collection.csv gpurun_out/pmc_w/w_counter_collection.csv profiles/r02_pmc_traffic.json [train steps in the run = 3] [commit]

Counter unit is KB.  FETCH_SIZE is doubled: on gfx950 it reports 1/2 of the bytes of a wide coalesced (16 B/lane)
read (MI355X_MICROARCH.md, HBM section); all kernels here read with 16-byte pieces.  WRITE_SIZE is taken as is."""
import collections
import csv
import json
import re
import sys


def pretty(k):
    """Kernel name as written in the sources.  rocprofv3 demangles most names; those with a bf16 template argument (DF16b, which
    its demangler does not know) arrive mangled: _ZN12_GLOBAL__N_1<len><name>I<literal args>E... -> name<32, 2, true, bf16>."""
    m = re.match(r"_ZN12_GLOBAL__N_1(\d+)", k)
    if m:
        n = int(m.group(1))
        start = m.end()
        name, rest = k[start:start + n], k[start + n:]
        args = []
        if rest.startswith("I"):
            i = 1
            while i < len(rest) and rest[i] != "E":
                t = re.match(r"L([ib])(n?\d+)E", rest[i:])
                if t:
                    v = t.group(2).replace("n", "-")
                    args.append(v if t.group(1) == "i" else ("true" if v == "1" else "false"))
                    i += t.end()
                elif rest.startswith("DF16b", i):
                    args.append("bf16"); i += 5
                elif rest[i] == "f":
                    args.append("float"); i += 1
                else:
                    break
        return name + ("<" + ", ".join(args) + ">" if args else "")
    name = re.sub(r"\(anonymous namespace\)::", "", k)
    return re.sub(r"\(.*", "", name).replace("void ", "")


def load(path, counter):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            d[r["Kernel_Name"]].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    return d


def main(fcsv, wcsv, out, steps="3", commit="?"):
    steps = int(steps)
    f, w = load(fcsv, "FETCH_SIZE"), load(wcsv, "WRITE_SIZE")
    rows = []
    for k in f:
        if "anonymous" not in k and not k.startswith("_ZN12_GLOBAL__N_1"):
            continue
        name = pretty(k)
        fv, wv, dur = [x[0] for x in f[k]], [x[0] for x in w.get(k, [(0, 0)])], [x[1] for x in f[k]]
        fetch, write = sum(fv) / len(fv) * 1024 * 2, sum(wv) / len(wv) * 1024
        rows.append(dict(kernel=name, launches=len(fv), launches_per_step=len(fv) / steps, avg_us_profiled=sum(dur) / len(dur) / 1e3,
                         fetch_bytes_corrected=fetch, write_bytes=write, hbm_bytes_per_launch=fetch + write))
    rows.sort(key=lambda r: -r["avg_us_profiled"] * r["launches"])
    json.dump(dict(note=__doc__, source_commit=commit, train_steps_in_run=steps, kernels=rows), open(out, "w"), indent=1)
    print("%-46s %4s %9s %10s %10s %8s" % ("kernel", "n", "avg us", "fetch MB", "write MB", "GB/s"))
    for r in rows[:30]:
        print("%-46s %4d %9.1f %10.1f %10.1f %8.0f" % (r["kernel"][:46], r["launches"], r["avg_us_profiled"], r["fetch_bytes_corrected"] / 1e6,
                                                      r["write_bytes"] / 1e6, r["hbm_bytes_per_launch"] / r["avg_us_profiled"] / 1e3))


if __name__ == "__main__":
    main(*sys.argv[1:6])
