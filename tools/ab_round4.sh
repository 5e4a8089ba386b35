#!/bin/bash
# A/B pairs of round 4 whose raw lines go to profiles/r04_ab.md (run on the GPU box through gpurun, from the repo root; ~1 minute):
#   1. host step, narrow wire format: pose / reward / done / info stored straight into the caller's block (default) vs staged + copied
#   2. fused rollout: prepared next episodes (default: one launch per refill interval) vs ONE launch with inline resets
#   3. host step: delta refresh + narrowing as one launch (default) vs two
# Each pair alternated twice on the same box.
OUT=gpurun_out/r04; mkdir -p $OUT
{ for i in 1 2; do for d in 1 0; do
    echo "NGW_WIRE_DIRECT=$d"
    NGW_WIRE_DIRECT=$d timeout -k 10 120 python tools/api_mode_rate.py 2>&1 | grep "API mode"
    NGW_WIRE_DIRECT=$d timeout -k 10 120 python tools/api_latency.py 2>&1 | grep "n = "
  done; done; } > $OUT/ab_wire_direct.log 2>&1 || exit 1
{ for i in 1 2; do for p in auto 0; do
    echo "bench.py --mode rollout --reset-prefetch $p"
    timeout -k 10 200 python bench.py --mode rollout --reset-prefetch $p --no-cpu-baseline --no-side 2>/dev/null | grep "^{" | tail -1 \
      | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  value %.2f G env-steps/s, ms_per_step %.5f, steps %d; %s' % (d['value']/1e9, d['ms_per_step'], d['steps'], d['config']['mode']))"
  done; done; } > $OUT/ab_rollout_modes.log 2>&1 || exit 1
{ for i in 1 2; do for d in 1 0; do
    echo "NGW_WIRE_MERGE=$d"
    NGW_WIRE_MERGE=$d timeout -k 10 120 python tools/api_mode_rate.py 2>&1 | grep "API mode"
    NGW_WIRE_MERGE=$d timeout -k 10 120 python tools/api_latency.py 2>&1 | grep "n = "
  done; done; } > $OUT/ab_wire_merge.log 2>&1 || exit 1
cat $OUT/ab_wire_direct.log $OUT/ab_rollout_modes.log $OUT/ab_wire_merge.log
