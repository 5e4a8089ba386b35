#!/bin/bash
# Profiling recipe of a round (run on the GPU box through gpurun, from the repo root):
#   1. rocprofv3 --kernel-trace --stats of the default bench command (step mode) and of the fused rollout
#   2. HBM traffic counters in SEPARATE passes (FETCH_SIZE, then WRITE_SIZE; TCC slots do not fit both) for both modes
#   4. kernel traces of the staggered-episode-ends regime, inline resets vs prepared next episodes
#   3. the same counters on the staging-only diagnostic kernel at 1 Mi envs (working set > Infinity Cache) whose byte
#      count is known exactly: calibrates FETCH_SIZE (gfx950 reports 1/2 of wide coalesced reads, MI355X_MICROARCH.md §HBM)
# Outputs land in gpurun_out/prof_$TAG/ ; tools/parse_profiles.py turns them into profiles/*.md + pmc_traffic.json.
set -e
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
rm -rf $OUT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python3 bench.py --no-cpu-baseline --no-stagger --steps 400 --warmup 100"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_step -- $B > $OUT/stats_step.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_rollout -- $B --mode rollout > $OUT/stats_rollout.log 2>&1
P="python3 bench.py --no-cpu-baseline --no-stagger --steps 60 --warmup 10 --launch eager"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_step_$c -- $P > $OUT/pmc_step_$c.log 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_rollout_$c -- $P --mode rollout > $OUT/pmc_rollout_$c.log 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_calib_$c -- python3 tools/dbg_launch.py calib > $OUT/pmc_calib_$c.log 2>&1
done
# 4. staggered episode ends: inline resets vs prepared next episodes (ngw_set_reset_prefetch), kernel trace of each
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_stagger_inline -- python3 tools/stagger_rate.py > $OUT/stats_stagger_inline.log 2>&1
export NGW_PREFETCH=32
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_stagger_prefetch -- python3 tools/stagger_rate.py > $OUT/stats_stagger_prefetch.log 2>&1
unset NGW_PREFETCH
echo profile_round done
