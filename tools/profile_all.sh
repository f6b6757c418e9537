#!/bin/bash
# Reproduces everything under profiles/ for one build:  bash tools/profile_all.sh <tag>     (run ON the MI355X box, repo root)
#   1. (runs LAST, so that its roofline.traffic comes from this build's own PMC summary) un-profiled bench line -> gpurun_out/<tag>_bench_4096.json
#   2. rocprofv3 --kernel-trace --stats of the bench   -> gpurun_out/<tag>_stats/
#   3. three separate --pmc passes (SQ set, FETCH_SIZE, WRITE_SIZE) at 262144 envs and FETCH/WRITE at 4096 envs
#   4. tools/pmc_summary.py                            -> gpurun_out/<tag>_pmc_summary.json, <tag>_kernel_stats_bench_4096.csv
#   5. kernel trace of VecTask.step() calls            -> gpurun_out/<tag>_step_api_kernel_stats.csv
#   5b. kernel trace of taco_rollout_run                -> gpurun_out/<tag>_rollout_kernel_stats.csv
#   5c. SQ counters of the rollout's kernels (2 passes) -> gpurun_out/<tag>_rollout_pmc.json (tools/pmc_kernels.py)
#   5d. size-isolated kernel-trace tables (262 144 / 1 M envs, rotate / flip 16 384, mix 32 768 x 5 and 262 144 x 5, the critic at 557 056 / 135 168 rows,
#       the rollout at 32 768 x 16)                     -> gpurun_out/<tag>_kernel_stats_<shape>.csv
#   6. bench.py --gpus 2 / --gpus 4 over gloo on this one GPU -> gpurun_out/<tag>_bench_{2,4}ranks_gloo_one_gpu.json
#   8. tools/cell_ab.py, tools/fresh_outputs_cost.py    -> gpurun_out/<tag>_cell_ab.txt, <tag>_fresh_outputs_cost.txt
# --pmc is never combined with any trace domain other than --kernel-trace; python3 is the program right after `--`.
set -eo pipefail
TAG=${1:-r01_x}
R=$(pwd)
O=$R/gpurun_out
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/${TAG}_stats" -- python3 "$R/bench.py" --steps 2000 --warmup 200 --no-cpu-baseline > "$O/${TAG}_bench_prof.json" 2> "$O/${TAG}_bench_prof.err"
SQ="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_WAIT_ANY"
rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d "$O/${TAG}_pmc_sq_262144" -- python3 "$R/tools/prof_step.py" --envs 262144 --steps 700 > "$O/${TAG}_pmc.log" 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$O/${TAG}_pmc_${c}_262144" -- python3 "$R/tools/prof_step.py" --envs 262144 --steps 700 >> "$O/${TAG}_pmc.log" 2>&1
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$O/${TAG}_pmc_${c}_4096" -- python3 "$R/tools/prof_step.py" --envs 4096 --steps 700 >> "$O/${TAG}_pmc.log" 2>&1
done
# 3a. the headline launch's own instruction counts (4 096 envs, the form the product picks, steady state): bench.py's roofline.valu
rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d "$O/${TAG}_pmc_sq_4096" -- python3 "$R/tools/prof_step.py" --envs 4096 --steps 700 >> "$O/${TAG}_pmc.log" 2>&1
# 3b. instructions of ONE wavefront that runs a whole step (the quad form without role wavefronts: every wavefront is a step wavefront) at 4 096 envs:
#     the count behind bench.py's latency_floor
rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d "$O/${TAG}_pmc_sq_4096_quad" -- python3 "$R/tools/prof_step.py" --envs 4096 --steps 40 --form quad >> "$O/${TAG}_pmc.log" 2>&1
# 5. VecTask.step() issues ONE kernel per call: kernel trace of 200 step() calls at 4 096 envs (copy_outputs=True: clamped copies included)
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/${TAG}_step_api" -- python3 "$R/tools/prof_step.py" --api --envs 4096 --steps 200 >> "$O/${TAG}_pmc.log" 2>&1
# 5b. one PPO rollout (config 5's flags, 4 096 envs x 32 steps): the persistent actor + step kernel, the batched critic, GAE
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/${TAG}_rollout" -- python3 "$R/tools/prof_rollout.py" >> "$O/${TAG}_pmc.log" 2>&1
# 5c. where the rollout's kernels spend their SIMD time: MFMA vs VALU instructions, busy cycles, LDS
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_INSTS_LDS --output-format csv -d "$O/${TAG}_rollout_pmc/a" -- python3 "$R/tools/prof_rollout.py" >> "$O/${TAG}_pmc.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$O/${TAG}_rollout_pmc/b" -- python3 "$R/tools/prof_rollout.py" >> "$O/${TAG}_pmc.log" 2>&1
# 5d. (round 6) SIZE-ISOLATED kernel-trace rows: every figure bench.py's large_n[] / configs[] quotes has a CSV whose AverageNs it can be checked against
#     (the bench run's own table lumps all launches of an instantiation: 262 144- and 1 M-env launches of <64,1,false,true,...> share one row)
iso() { # iso <name> <program> <args...>
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d "$O/${TAG}_iso_${name}" -- python3 "$@" >> "$O/${TAG}_pmc.log" 2>&1
  cp "$O/${TAG}_iso_${name}"/*/*kernel_stats.csv "$O/${TAG}_kernel_stats_${name}.csv"
}
iso 262144 "$R/tools/prof_step.py" --envs 262144 --steps 3000
iso 1048576 "$R/tools/prof_step.py" --envs 1048576 --steps 1000
iso rotate_16384 "$R/tools/prof_step.py" --config 2 --envs 16384 --steps 6000
iso flip_16384 "$R/tools/prof_step.py" --config 3 --envs 16384 --steps 6000
iso mix_32768x5 "$R/tools/prof_step.py" --config 4 --envs 32768 --steps 6000 --api
iso mix_262144x5 "$R/tools/prof_step.py" --config 4 --envs 262144 --steps 2000 --api
iso critic_557056 "$R/tools/prof_critic.py" --slots=17 --envs=32768 --reps=60
iso critic_135168 "$R/tools/prof_critic.py" --slots=33 --envs=4096 --reps=200
iso rollout_32768x16 "$R/tools/prof_rollout.py" --envs=32768 --horizon=16 --reps=20
cd "$R"
python3 tools/pmc_kernels.py "$O/${TAG}_rollout_pmc" "$O/${TAG}_rollout_pmc.json" taco_rollout_kernel taco_critic_lstm_pair_split_kernel taco_critic_lstm_pair_kernel taco_critic_mlp > /dev/null
python3 tools/pmc_summary.py "$TAG"
cp "$O/${TAG}_rollout"/*/*kernel_stats.csv "$O/${TAG}_rollout_kernel_stats.csv"
cp "$O/${TAG}_step_api"/*/*kernel_stats.csv "$O/${TAG}_step_api_kernel_stats.csv"
# 1. the un-profiled bench line, with this build's PMC summary where bench.py looks for it (profiles/, newest matching source hash)
cp "$O/${TAG}_pmc_summary.json" "$O/${TAG}"_kernel_stats_*.csv profiles/   # (the size-isolated tables too: bench.py names them beside its figures, rocprof_row)
python3 bench.py --steps 2000 --warmup 200 > "$O/${TAG}_bench_4096.json" 2> "$O/${TAG}_bench.err"
# 7. eager (clock in the kernel arguments / on the device) vs HIP-graph replays of 1, 16, 64 steps: what the device-resident clock costs
python3 tools/clock_path_ab.py > "$O/${TAG}_clock_path_ab.txt" 2>/dev/null || true
# 6. the N > 1 code path end to end on this one GPU: 2 ranks over gloo (the driver runs the real thing over RCCL on 8 GPUs)
TACO_BENCH_BACKEND=gloo TACO_BENCH_ONE_DEVICE=1 python3 bench.py --gpus 2 --steps 500 --warmup 100 --no-cpu-baseline --no-large-n --no-configs > "$O/${TAG}_bench_2ranks_gloo_one_gpu.json" 2> "$O/${TAG}_bench_2ranks.err" || true
# 6b. ... and with FOUR ranks incl. the BASELINE-total legs (the pool's process guard allows six processes on a card: four ranks + the launcher fit, six ranks do not)
TACO_BENCH_BACKEND=gloo TACO_BENCH_ONE_DEVICE=1 timeout -k 10 900 python3 bench.py --gpus 4 --steps 50 --warmup 10 --no-cpu-baseline --no-large-n > "$O/${TAG}_bench_4ranks_gloo_one_gpu.json" 2> "$O/${TAG}_bench_4ranks.err" || true
# 8. the batched critic: exact cell / hardware cell / split f16 / split bf16 (ms per values_ring, value differences), and what fresh_outputs costs
python3 tools/cell_ab.py > "$O/${TAG}_cell_ab.txt" 2>&1 || true
python3 tools/fresh_outputs_cost.py > "$O/${TAG}_fresh_outputs_cost.txt" 2>&1 || true
