#!/bin/bash
# usage (on the GPU box): bash tools/collect_config.sh <tag> <commit> <bench args...>
# kernel stats + PMC HBM traffic (separate FETCH_SIZE / WRITE_SIZE passes) of one bench.py configuration, under gpurun_out/<tag>/
set -e
T=$1; COMMIT=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$T
mkdir -p $O
python3 bench.py "$@" > $O/bench.json 2> $O/bench.err
python3 bench.py "$@" --breakdown --no-cpu-baseline --no-exact-line > $O/bench_breakdown.json 2> $O/breakdown.txt
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $O/ks -o r --output-format csv -- python3 bench.py "$@" --steps 5 --warmup 2 --no-cpu-baseline --no-fwd-eval --no-exact-line > $O/ks.log 2>&1
timeout -k 10 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_f -o f --output-format csv -- python3 bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-fwd-eval --no-exact-line > $O/pmc_f.log 2>&1
timeout -k 10 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_w -o w --output-format csv -- python3 bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-fwd-eval --no-exact-line > $O/pmc_w.log 2>&1
python3 profiles/collect_pmc.py $O/pmc_f/f_counter_collection.csv $O/pmc_w/w_counter_collection.csv $O/pmc_traffic.json 3 $COMMIT > $O/pmc_traffic.txt
cp $(find $O/ks -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
echo collected $T
