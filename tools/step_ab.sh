#!/bin/bash
# Step-kernel tuning call: stamps timeline + C2/C4 bench (1000-step graph and the driver's 20-step eager form), optional env A/B
OUT=gpurun_out/${ROUND:-r04}; mkdir -p $OUT
bash tools/stamps.sh -- C2 2>&1 | tail -22
run() { # label, env...
  local L=$1; shift
  for W in C2 C4; do
    env "$@" timeout -k 10 200 python bench.py --workload $W --no-cpu-baseline --no-side --steps 1000 > $OUT/ab_${L}_$W.log 2>&1
    env "$@" timeout -k 10 200 python bench.py --workload $W --no-cpu-baseline --no-side --steps 20 --warmup 5 > $OUT/ab_${L}_${W}_20.log 2>&1
    python - <<PY
import json
def g(f):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1]); return "%.2f G %.2f us frac %.3f" % (d["value"]/1e9, d["ms_per_step"]*1000, d["roofline"]["frac"])
    except Exception as ex: return "FAILED %s" % ex
print("$L $W  1000-step graph: %s | 20-step eager: %s" % (g("$OUT/ab_${L}_$W.log"), g("$OUT/ab_${L}_${W}_20.log")))
PY
  done
}
run base NGW_DUMMY=1

