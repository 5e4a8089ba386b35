#!/bin/bash
# bench lines of the named workloads (no CPU baseline, sides on): tools/bench_lines.sh C5 X1 ...
OUT=gpurun_out/${ROUND:-r04}; mkdir -p $OUT
for W in "$@"; do
  timeout -k 10 400 python bench.py --workload $W --no-cpu-baseline --steps 1000 > $OUT/bench_$W.log 2>&1
  python - <<PY
import json
try:
    d = json.loads([l for l in open("$OUT/bench_$W.log") if l.startswith("{")][-1])
    st = d.get("staggered_resets", {})
    pe = st.get("prepared_next_episodes", {})
    print("$W %.2f G  %.2f us/step  frac %.3f resets %d prepared %s | staggered: inline %.1f us, prepared %.1f us" % (d["value"] / 1e9, d["ms_per_step"] * 1000, d["roofline"]["frac"], d["resets_in_timed_region"], d.get("prepared_episodes"), st.get("inline_resets", {}).get("ms_per_step", 0) * 1e3, pe.get("ms_per_step", 0) * 1e3))
except Exception as ex:
    print("$W FAILED", ex); print(open("$OUT/bench_$W.log").read()[-1500:])
PY
done
