#!/bin/bash
# usage (on the GPU box): bash tools/collect_round.sh <tag> <commit>  -> everything the round's profiles/ entries are made from,
# under gpurun_out/<tag>_*/ (copy what is to be judged into profiles/ afterwards; profiles/README.md names the files)
set -e
T=$1
COMMIT=${2:-unknown}
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out      # (every path below is relative to the repo root)
bash tools/collect_config.sh ${T}_f32 $COMMIT                      # the default line (--math f32 = exact f32, the `f22` side line beside it)
bash tools/collect_config.sh ${T}_f22 $COMMIT --math f22           # f16 pieces (exact_f32 beside it)
bash tools/pmc_sq.sh --math f32 > gpurun_out/${T}_sq_f32.txt 2> gpurun_out/${T}_sq.err
bash tools/pmc_sq.sh --math f22 > gpurun_out/${T}_sq_f22.txt 2>> gpurun_out/${T}_sq.err
cd $GRAFT_REPO_ROOT
O=gpurun_out/${T}_lines
mkdir -p $O
python3 bench.py --math split --no-cpu-baseline --breakdown > $O/bench_split.json 2> $O/bench_split.txt
python3 bench.py --math fwd22 --no-cpu-baseline --breakdown > $O/bench_fwd22.json 2> $O/bench_fwd22.txt
python3 bench.py --config 5 > $O/bench_config5_cnn3d.json 2> $O/bench_config5.err
python3 bench.py --config 4 --steps 10 --warmup 3 > $O/bench_config4_f32.json 2> $O/bench_config4_f32.err
python3 bench.py --config 4 --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_config4_bf16.json 2> $O/bench_config4_bf16.err
python3 bench.py --config 1 --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_config1.json 2> $O/bench_config1.err
IDEE_DIST_BACKEND=gloo IDEE_FORCE_DEVICE=0 python3 bench.py --gpus 2 --steps 10 --warmup 3 > $O/bench_gloo2_rehearsal.json 2> $O/bench_gloo2.err
echo collected
