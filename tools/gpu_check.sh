#!/bin/bash
# One GPU-box call while tuning: parity tests, in-kernel timeline of the stamps build, A/B bench lean vs general kernel.
#   tools/gpu_check.sh [tests|notests] [workloads...]
OUT=gpurun_out/r02; mkdir -p $OUT
MODE=${1:-tests}; shift
WL=${@:-C2}
if [ "$MODE" = tests ]; then
  timeout -k 10 900 python -m pytest tests -x -q -m gpu > $OUT/gputest.log 2>&1; tail -4 $OUT/gputest.log
fi
if [ -f gym_novel_gridworlds_amd/libngw_hip_stamps.so ]; then
  NGW_LIB=$PWD/gym_novel_gridworlds_amd/libngw_hip_stamps.so timeout -k 10 300 python tools/stamp_timeline.py $WL > $OUT/stamps.log 2>&1; grep -v amdgpu.ids $OUT/stamps.log
fi
for W in $WL; do
  for L in 1 0; do
    NGW_LEAN=$L timeout -k 10 200 python bench.py --workload $W --no-cpu-baseline --no-side --steps 1000 > $OUT/bench_${W}_lean$L.log 2>&1
    python - <<PY
import json
try:
    d = json.loads([l for l in open("$OUT/bench_${W}_lean$L.log") if l.startswith("{")][-1])
    print("$W lean=$L %.2f G  %.2f us/step  frac %.3f resets %d" % (d["value"] / 1e9, d["ms_per_step"] * 1000, d["roofline"]["frac"], d["resets_in_timed_region"]))
except Exception as ex:
    print("$W lean=$L FAILED", ex); print(open("$OUT/bench_${W}_lean$L.log").read()[-1500:])
PY
  done
done
