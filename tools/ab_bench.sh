#!/bin/bash
# A/B two builds of libngw_hip.so in ONE gpurun call (box-to-box variance is ~5 %): interleaved bench runs.
A=${1:-gym_novel_gridworlds_amd/libngw_hip.so}; B=${2:-gym_novel_gridworlds_amd/libngw_hip_prev.so}
for rep in 1 2 3; do
  for lib in $A $B; do
    NGW_LIB=$PWD/$lib python bench.py --no-cpu-baseline --no-stagger --steps 600 2>/dev/null | python -c "
import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$lib', round(d['value']/1e9,2), 'G', d['ms_per_step']*1000, 'us | fused', round(d['fused_rollout']['value']/1e9,2), 'G')"
  done
done
