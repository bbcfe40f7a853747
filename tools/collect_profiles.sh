#!/bin/bash
# Collects the rocprofv3 evidence kept under profiles/ (run on the GPU box through gpurun; raw output goes to gpurun_out/).
#   bash tools/collect_profiles.sh <tag>
# Passes (each its own run; counters never share a run with tracing beyond --kernel-trace):
#   1. --kernel-trace --stats of the bench command (no child processes: the CPU-baseline and shim legs spawn programs)
#   1b. the same of the driver's short run (--steps 20 --warmup 5)
#   2. --pmc FETCH_SIZE   3. --pmc WRITE_SIZE   of the same command with fewer steps; 2b / 3b: of the headline alone
#   4. --pmc SQ_* of the dense-kernel microbenchmark (issue-bound evidence) + its plain output (A/B table)
#   5. the VALU instruction-rate microbenchmark
set -u
TAG=${1:-r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
rm -rf $OUT   # (gpurun MERGES what a call writes into the caller's gpurun_out/: use a fresh tag per collection there, or delete
mkdir -p $OUT  #  the local copy first — rocprofv3 names its files by PID, and PIDs repeat from box to box)
cd /tmp && export TMPDIR=/tmp
# Under rocprofv3 every launch costs the host several times what it costs unprofiled, so the tracker's depth worker always finds its
# next job already waiting and (tracker_worker_main's rule for jobs that start in the middle of a frame) would send nearly all of
# them down the launch-per-iteration path — 4 % of the jobs unprofiled (ODO_LOG_GIVEUPS=1 prints the split at exit). The profiled runs
# keep the unprofiled run's path:
export ODO_DEPTH_PERSIST_ALWAYS=1
BENCH="python3 $GRAFT_REPO_ROOT/bench.py --cpu-frames 0 --no-child-processes --details $OUT/bench_profiled_details.json --extras dense,disparity,single,batched,saturated"
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH --steps 1000 --warmup 50 > $OUT/bench_profiled.json 2> $OUT/bench_profiled.err < /dev/null
# 1b. the driver's own short run (python bench.py --steps 20 --warmup 5), side legs off: the averages bench.py's roofline line must agree with
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats20 -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-frames 0 --no-extras --details $OUT/bench_profiled_steps20_details.json --steps 20 --warmup 5 > $OUT/bench_profiled_steps20.json 2> $OUT/bench_profiled_steps20.err < /dev/null
timeout -k 5 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH --steps 199 --warmup 5 > /dev/null 2>&1 < /dev/null
timeout -k 5 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH --steps 199 --warmup 5 > /dev/null 2>&1 < /dev/null
# 2b / 3b. the same two counters of the HEADLINE ALONE (no side legs, no stress drive): lm_fine_kernel's traffic per launch on the very
# kernel configuration bench.py's roofline line is printed for (the mixed passes above average three legs' launches)
HEAD="python3 $GRAFT_REPO_ROOT/bench.py --cpu-frames 0 --no-extras --no-stress --details /tmp/head_details.json --steps 199 --warmup 5"
timeout -k 5 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_headline -- $HEAD > /dev/null 2>&1 < /dev/null
timeout -k 5 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_headline -- $HEAD > /dev/null 2>&1 < /dev/null
F="-O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fhip-fp32-correctly-rounded-divide-sqrt"
hipcc $F -o /tmp/dense_ablate $GRAFT_REPO_ROOT/tools/microbench/dense_ablate.hip 2> $OUT/build.log
timeout -k 5 120 /tmp/dense_ablate > $OUT/dense_ablate.log 2>&1 < /dev/null
timeout -k 5 120 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -- /tmp/dense_ablate 1920 1080 "pipelined, shared reciprocals [1024" > /dev/null 2>&1 < /dev/null
timeout -k 5 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_ablate -- /tmp/dense_ablate 1920 1080 "1024 x 256" > /dev/null 2>&1 < /dev/null
hipcc -O3 --offload-arch=gfx950 -o /tmp/valu_rates $GRAFT_REPO_ROOT/tools/microbench/valu_rates.hip 2>> $OUT/build.log
timeout -k 5 120 /tmp/valu_rates > $OUT/valu_rates.log 2>&1 < /dev/null
# 6. (round 6, VERDICT r05 task 2) the dense 1080p leg alone — configs[2] — with its per-dispatch trace kept, so that level 0 of
#    lm_dense_eval_kernel is its own row (tools/summarize_profile.py grids: rows per kernel AND grid size), and one SQ pass of the
#    same command; 7. the SQ pass of the headline (the LM chain's wait / issue split)
DENSE="python3 $GRAFT_REPO_ROOT/bench.py --cpu-frames 0 --no-child-processes --no-causal --extras dense --no-stress --details $OUT/bench_dense_details.json --steps 20 --warmup 5"
SQ="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_dense -- $DENSE > $OUT/bench_dense.json 2> $OUT/bench_dense.err < /dev/null
for f in $(find $OUT/stats_dense -name "*_kernel_trace.csv"); do mv $f ${f%_kernel_trace.csv}_dispatches.csv; done
timeout -k 5 300 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $OUT/pmc_sq_dense -- $DENSE > /dev/null 2>&1 < /dev/null
timeout -k 5 300 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $OUT/pmc_sq_headline -- $HEAD --no-causal > /dev/null 2>&1 < /dev/null
# keep the per-kernel stats and the counter collections (what the summaries are made from); drop the per-dispatch traces
find $OUT -name "*_kernel_trace.csv" -delete
find $OUT -name "*_agent_info.csv" -delete
du -sh $OUT; ls -R $OUT | head -60
