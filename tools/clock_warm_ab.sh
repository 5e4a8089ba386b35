for cw in 0 250 0 250 0 250; do
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-side --clock-warm-ms $cw > gpurun_out/cw.log 2>&1
  python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/cw.log") if l.startswith("{")][-1])
print("clock-warm $cw ms: %.2f G  %.2f us/step  frac %.3f  kernel avg %.2f us" % (d["value"]/1e9, d["ms_per_step"]*1000, d["roofline"]["frac"], d["roofline"].get("launch_period_ms_event_pair", d["roofline"].get("kernel_ms_avg", 0))*1000))
PY
done
