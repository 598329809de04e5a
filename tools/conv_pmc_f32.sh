#!/bin/bash
# usage (on the GPU box): bash tools/conv_pmc.sh <tag> <conv_bench args...> — timing + SQ counters of the isolated f32-storage conv passes (tools/conv_bench_f32.py)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; shift
mkdir -p $O
python3 tools/conv_bench_f32.py "$@" | tee $O/time.txt
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT -d $O/p1 -o s --output-format csv -- python3 tools/conv_bench_f32.py "$@" --reps 2 > $O/p1.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE -d $O/p2 -o s --output-format csv -- python3 tools/conv_bench_f32.py "$@" --reps 2 > $O/p2.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_ACTIVE_INST_FLAT -d $O/p3 -o s --output-format csv -- python3 tools/conv_bench_f32.py "$@" --reps 2 > $O/p3.log 2>&1
python3 - $O <<'PY'
import csv, collections, re, sys
sys.path.insert(0, "profiles")
from collect_pmc import pretty
O = sys.argv[1]
for d in ("p1", "p2", "p3"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
    try: rows = list(csv.DictReader(open(f"{O}/{d}/s_counter_collection.csv")))
    except Exception as e: print(d, "missing", e); continue
    for r in rows:
        k = pretty(r["Kernel_Name"])
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"])); dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    names = sorted({c for k in acc for c in acc[k]})
    print("kernel".ljust(34), "us".rjust(8), " ".join(n.replace("SQ_", "")[:14].rjust(14) for n in names))
    for k, c in sorted(acc.items(), key=lambda kv: -sum(dur[kv[0]]))[:6]:
        print(k[:34].ljust(34), ("%.0f" % (sum(dur[k]) / len(dur[k]) / 1e3)).rjust(8),
              " ".join(("%.3g" % (sum(c[nm]) / len(c[nm]))).rjust(14) if nm in c else "-".rjust(14) for nm in names))
PY
