#!/bin/bash
# usage (on the GPU box): bash tools/collect_round.sh <tag>   -> everything the round's profiles/ entries are made from, under gpurun_out/<tag>/
set -e
T=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$T
mkdir -p $O
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
python3 bench.py --no-cpu-baseline --breakdown > $O/bench_breakdown.json 2> $O/bench_breakdown.txt
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $O/ks -o r --output-format csv -- python3 bench.py --no-cpu-baseline > $O/ks.log 2>&1
timeout -k 10 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_f -o f --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_f.log 2>&1
timeout -k 10 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_w -o w --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_w.log 2>&1
python3 profiles/collect_pmc.py $O/pmc_f/f_counter_collection.csv $O/pmc_w/w_counter_collection.csv $O/pmc_traffic.json > $O/pmc_traffic.txt
python3 bench.py --no-cpu-baseline --encoder CNN_3D > $O/bench_cnn.json 2> $O/bench_cnn.err
echo collected
