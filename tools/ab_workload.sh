#!/bin/bash
# A/B two builds of libngw_hip.so on another workload in ONE gpurun call: tools/ab_workload.sh C5 [steps]
W=${1:-C5}; K=${2:-300}
A=gym_novel_gridworlds_amd/libngw_hip.so; B=gym_novel_gridworlds_amd/libngw_hip_prev.so
for rep in 1 2; do
  for lib in $A $B; do
    NGW_LIB=$PWD/$lib python bench.py --no-cpu-baseline --no-stagger --workload $W --steps $K 2>/dev/null | python -c "
import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$lib', '$W', round(d['value']/1e9,3), 'G', round(d['ms_per_step']*1000,2), 'us | fused', round(d['fused_rollout']['value']/1e9,3), 'G')"
  done
done
