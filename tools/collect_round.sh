#!/bin/bash
# usage (on the GPU box): bash tools/collect_round.sh <tag> <commit>  -> everything the round's profiles/ entries are made from,
# under gpurun_out/<tag>/ (copy what is to be judged into profiles/ afterwards)
set -e
T=$1
COMMIT=${2:-unknown}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$T
mkdir -p $O
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
python3 bench.py --no-cpu-baseline --breakdown > $O/bench_breakdown.json 2> $O/bench_breakdown.txt
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $O/ks -o r --output-format csv -- python3 bench.py --no-cpu-baseline --no-fwd-eval > $O/ks.log 2>&1
timeout -k 10 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_f -o f --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fwd-eval > $O/pmc_f.log 2>&1
timeout -k 10 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_w -o w --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fwd-eval > $O/pmc_w.log 2>&1
python3 profiles/collect_pmc.py $O/pmc_f/f_counter_collection.csv $O/pmc_w/w_counter_collection.csv $O/pmc_traffic.json 3 $COMMIT > $O/pmc_traffic.txt
python3 bench.py --no-cpu-baseline --math split --breakdown > $O/bench_split.json 2> $O/bench_split.txt
python3 bench.py --config 5 > $O/bench_config5_cnn3d.json 2> $O/bench_config5.err
python3 bench.py --config 4 --steps 10 --warmup 3 > $O/bench_config4_era5.json 2> $O/bench_config4.err
python3 bench.py --config 1 --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_config1.json 2> $O/bench_config1.err
IDEE_DIST_BACKEND=gloo IDEE_FORCE_DEVICE=0 python3 bench.py --gpus 2 --steps 10 --warmup 3 > $O/bench_gloo2_rehearsal.json 2> $O/bench_gloo2.err
echo collected
