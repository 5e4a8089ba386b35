#!/bin/bash
# 10 x 10: the step kernel that stages the maps through LDS against the one that reads them in place, per batch size (1000-step graph replay).
OUT=gpurun_out/${ROUND:-r04}; mkdir -p $OUT
for W in C2 C4 C2@16384 C2@4096 C2@64; do
  ENVS=""; case $W in *@*) ENVS="--envs ${W#*@}";; esac
  WL=${W%@*}
  for NS in 0 100; do
    export NGW_NOSTAGE=$NS      # 0: staged at every size the LDS takes; 100: in place from 10 x 10 up
    timeout -k 10 200 python bench.py --workload $WL $ENVS --steps 1000 --warmup 100 --no-side --no-cpu-baseline --repeats 3 > $OUT/ab_stage_${W}_$NS.json 2> $OUT/ab_stage_${W}_$NS.err
    python - <<PY
import json
d = json.load(open('$OUT/ab_stage_${W}_$NS.json'))
print('$W NGW_NOSTAGE=$NS: %s  value %.2f G  ms_per_step %.4f  device %.4f  repeats device median %.4f' % (d['roofline']['kernel'], d['value'] / 1e9, d['ms_per_step'], d['roofline'].get('launch_period_ms_event_pair', d['roofline'].get('kernel_ms_avg', 0)), d['repeats']['ms_per_step_device']['median']))
PY
  done
done
unset NGW_NOSTAGE
