#!/bin/bash
# SQ instruction-mix counters of the fused rollout (per-wave, per-step averages): tools/sq_counters.sh
OUT=gpurun_out/sq
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/$tag -- python3 bench.py --no-cpu-baseline --no-stagger --mode ${NGW_SQ_MODE:-rollout} --launch eager --steps 200 --warmup 20 > $OUT/$tag.log 2>&1 || echo "set failed: $set"
done
python3 - <<'PY'
import csv, glob, re
rows = {}
for f in glob.glob('gpurun_out/sq/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r'ngw_kernel<\d+, (\d+),', r['Kernel_Name'])
        if m and m.group(1) == '2':
            rows.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
steps, waves = 200, 1024
for k, v in sorted(rows.items()):
    print('%-24s timed launch total %.4g   per wave per step %.1f' % (k, v[-1], v[-1] / waves / steps))
PY
