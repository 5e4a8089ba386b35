#!/bin/bash
# One-call performance matrix (all BASELINE workloads, lidar, staggered resets, reset launch): run after any kernel or LDS
# layout change - an extra 8 KB of LDS once cost C5 and the fused lidar path half their speed without touching C2.
for w in C2 C3 C4 C5; do
  python bench.py --no-cpu-baseline --no-stagger --workload $w --steps 400 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$w  per-launch %.2f G (%.2f us, frac %.3f)  fused %.2f G' % (d['value']/1e9, d['ms_per_step']*1e3, d['roofline']['frac'], d['fused_rollout']['value']/1e9))"
done
python tools/rollout_actions_rate.py 2>/dev/null | tail -2
python tools/lidar_rate.py 2>/dev/null | tail -5
python tools/stagger_rate.py 2>/dev/null | grep staggered
NGW_PREFETCH=32 python tools/stagger_rate.py 2>/dev/null | grep staggered
bash tools/reset_trace.sh | head -3
python tools/api_latency.py 2>/dev/null | grep '^n ='
python tools/adapter_latency.py 2>/dev/null | grep adapter
