#!/bin/bash
# The driver's own command, five times (box-to-box and run-to-run spread of a 20-step region is ~3 %), + the default 1000-step line
OUT=gpurun_out/${ROUND:-r04}; mkdir -p $OUT
for i in 1 2 3 4 5; do
  timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-side > $OUT/driver_$i.log 2>&1
  python - <<PY
import json
d = json.loads([l for l in open("$OUT/driver_$i.log") if l.startswith("{")][-1])
print("driver form run $i: %.2f G  %.2f us/step  frac %.3f  kernel avg %.2f us" % (d["value"]/1e9, d["ms_per_step"]*1000, d["roofline"]["frac"], d["roofline"].get("launch_period_ms_event_pair", d["roofline"].get("kernel_ms_avg", 0))*1000))
PY
done
for W in "$@"; do
  timeout -k 10 200 python bench.py --workload $W --no-cpu-baseline --no-side --steps 1000 > $OUT/k1000_$W.log 2>&1
  python - <<PY
import json
d = json.loads([l for l in open("$OUT/k1000_$W.log") if l.startswith("{")][-1])
print("$W 1000 steps: %.2f G  %.2f us/step  frac %.3f (%s)" % (d["value"]/1e9, d["ms_per_step"]*1000, d["roofline"]["frac"], d["roofline"].get("bytes_model")))
PY
done
