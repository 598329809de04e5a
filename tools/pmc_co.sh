#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 500 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE -d gpurun_out/pmc_co -o s --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_co.log 2>&1
python3 - <<PY
import csv, collections, re
acc=collections.defaultdict(lambda: collections.defaultdict(list)); dur=collections.defaultdict(list)
for r in csv.DictReader(open("gpurun_out/pmc_co/s_counter_collection.csv")):
    k=re.sub(r"\(anonymous namespace\)::","",r["Kernel_Name"]); k=re.sub(r"\(.*","",k).replace("void ","")
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"])); dur[k].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
names=sorted({c for k in acc for c in acc[k]})
print("kernel".ljust(40), "us".rjust(8), " ".join(n.replace("SQ_","")[:16].rjust(16) for n in names))
for k,c in sorted(acc.items(), key=lambda kv:-sum(dur[kv[0]]))[:12]:
    print(k[:40].ljust(40), ("%.0f"%(sum(dur[k])/len(dur[k])/1e3)).rjust(8), " ".join(("%.3g"%(sum(c[nm])/len(c[nm]))).rjust(16) if nm in c else "-".rjust(16) for nm in names))
PY
