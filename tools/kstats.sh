#!/bin/bash
# usage: kstats.sh <tag> : rocprofv3 kernel stats of a short bench run, top kernels printed
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d gpurun_out/ks_$1 -o r --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/ks_$1.log 2>&1
python3 - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/ks_$1/r_kernel_stats.csv")))
for r in rows[:22]:
    n=r["Name"].replace("(anonymous namespace)::","").replace("void ","")
    print(n[:60].ljust(60), r["Calls"].rjust(5), ("%.3f"%(float(r["AverageNs"])/1e6)).rjust(9), r["Percentage"].rjust(7))
PY
grep -o '"value": [0-9.]*' gpurun_out/ks_$1.log
