#!/bin/bash
# usage (on the GPU box): bash tools/pmc_sq.sh [bench args...] > gpurun_out/<tag>_sq_counters.txt   (SQ counters per kernel, two passes)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 500 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT -d gpurun_out/pmc_sq -o s --output-format csv -- python3 bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-fwd-eval --no-exact-line > gpurun_out/pmc_sq.log 2>&1
timeout -k 10 500 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE -d gpurun_out/pmc_sq2 -o s --output-format csv -- python3 bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-fwd-eval --no-exact-line > gpurun_out/pmc_sq2.log 2>&1
python3 - <<PY
import csv, collections, re
for d in ("pmc_sq","pmc_sq2"):
    acc=collections.defaultdict(lambda: collections.defaultdict(list)); dur=collections.defaultdict(list)
    for r in csv.DictReader(open(f"gpurun_out/{d}/s_counter_collection.csv")):
        k=r["Kernel_Name"]
        m=re.match(r"_ZN12_GLOBAL__N_1(\d+)", k)
        if m: k=k[m.end():m.end()+int(m.group(1))]+"<"+re.sub(r"E+v?P.*","",k[m.end()+int(m.group(1)):]).replace("DF16b","bf16")+">"
        k=re.sub(r"\(anonymous namespace\)::","",k); k=re.sub(r"\(.*","",k).replace("void ","")
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"])); dur[k].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
    names=sorted({c for k in acc for c in acc[k]})
    print("kernel".ljust(40), "us".rjust(8), " ".join(n.replace("SQ_","")[:14].rjust(14) for n in names))
    rows=sorted(acc.items(), key=lambda kv:-sum(dur[kv[0]]))
    for k,c in rows[:14]:
        n=len(dur[k])//max(1,len(c))
        print(k[:40].ljust(40), ("%.0f"%(sum(dur[k])/len(dur[k])/1e3)).rjust(8), " ".join(("%.3g"%(sum(c[nm])/len(c[nm]))).rjust(14) if nm in c else "-".rjust(14) for nm in names))
PY
