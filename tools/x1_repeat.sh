for i in 1 2 3 4; do
  python3 bench.py --workload X1 --no-cpu-baseline --no-side > gpurun_out/x1_$i.log 2>&1
  python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/x1_$i.log") if l.startswith("{")][-1])
print("X1 run $i: %.2f G  %.2f us/step  prepared %s resets %d" % (d["value"]/1e9, d["ms_per_step"]*1000, d["prepared_episodes"], d["resets_in_timed_region"]))
PY
done
python3 tools/churn_probe.py X1 2>&1 | grep -v amdgpu | tail -12
