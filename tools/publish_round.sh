#!/bin/bash
# usage (in the repo, after `gpurun -- bash tools/collect_round.sh <tag> <commit>` merged gpurun_out/<tag>_* back):
#   bash tools/publish_round.sh <tag> <rNN>   -> copies the summaries profiles/README.md names into profiles/<rNN>_*
set -e
T=$1; R=$2; G=gpurun_out; P=profiles
cp $G/${T}_f32/bench.json            $P/${R}_bench_default.json
cp $G/${T}_f22/bench.json            $P/${R}_bench_f22.json
cp $G/${T}_f22/breakdown.txt         $P/${R}_bench_f22_breakdown.txt
cp $G/${T}_f22/kernel_stats.csv      $P/${R}_bench_f22_kernel_stats.csv
cp $G/${T}_f22/pmc_traffic.json      $P/${R}_pmc_traffic_f22.json
cp $G/${T}_f22/pmc_traffic.txt       $P/${R}_pmc_traffic_f22.txt
cp $G/${T}_sq_f22.txt                $P/${R}_sq_counters_f22.txt
cp $G/${T}_f32/breakdown.txt         $P/${R}_bench_f32_breakdown.txt
cp $G/${T}_f32/kernel_stats.csv      $P/${R}_bench_f32_kernel_stats.csv
cp $G/${T}_f32/pmc_traffic.json      $P/${R}_pmc_traffic.json
cp $G/${T}_f32/pmc_traffic.txt       $P/${R}_pmc_traffic.txt
cp $G/${T}_sq_f32.txt                $P/${R}_sq_counters.txt
L=$G/${T}_lines
cp $L/bench_split.json               $P/${R}_bench_split.json
cp $L/bench_split.txt                $P/${R}_bench_split_breakdown.txt
cp $L/bench_fwd22.json               $P/${R}_bench_fwd22.json
cp $L/bench_fwd22.txt                $P/${R}_bench_fwd22_breakdown.txt
cp $L/bench_config5_cnn3d.json       $P/${R}_bench_config5_cnn3d.json
cp $L/bench_config4_f32.json         $P/${R}_bench_config4_f32.json
cp $L/bench_config4_bf16.json        $P/${R}_bench_config4_bf16.json
cp $L/bench_config1.json             $P/${R}_bench_config1.json
cp $L/bench_gloo2_rehearsal.json     $P/${R}_bench_gloo2_rehearsal_one_gpu.json
echo published $T as $R
