#!/bin/bash
# usage (on the GPU box): bash tools/ks.sh <tag> <bench args...>   — kernel-trace stats only, top of the table printed
set -e
T=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$T
mkdir -p $O
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $O/ks -o r --output-format csv -- python3 bench.py "$@" --steps 5 --warmup 2 --no-cpu-baseline --no-fwd-eval > $O/ks.log 2>&1
cp $(find $O/ks -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
python3 - $O/kernel_stats.csv <<'PY'
import csv, sys
sys.path.insert(0, "profiles")
from collect_pmc import pretty
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:28]:
    print(f"{pretty(r['Name'])[:100]:100s} {int(r['Calls']):4d} {float(r['AverageNs'])/1e3:9.1f} us {float(r['Percentage']):6.2f} %")
PY
