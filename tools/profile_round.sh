#!/bin/bash
# Profiling recipe of a round: tools/profile_round.sh <round, e.g. r03>  (run on the GPU box through gpurun, from the repo root; ~7 minutes):
#   0. rocprofv3 --kernel-trace --stats of the DRIVER's command (python3 bench.py --gpus 1 --steps 20 --warmup 5)
#   1. rocprofv3 --kernel-trace --stats of the bench command, every workload (step mode) + the fused rollout at C2
#   2. HBM traffic counters in SEPARATE passes (FETCH_SIZE, then WRITE_SIZE: the TCC slots do not fit both), step mode of
#      every workload + the C2 rollout, and the same counters on the staging-only diagnostic kernel at 1 Mi envs whose byte
#      count is known exactly (calibrates FETCH_SIZE: gfx950 reports 1/2 of wide coalesced reads, MI355X_MICROARCH.md §HBM)
#   3. SQ counters (wave cycles, waiting, issue mix) of the C2 step kernel and of the C2 fused rollout
#   4. FETCH_SIZE / WRITE_SIZE of the new-episode kernel (explicit resets of every env, prepared episodes off): C3, C5, X1
#   5. the fused LidarInFront step: kernel trace per row format, FETCH / WRITE and SQ counters (int16 rows) -> profiles/<round>_lidar.md
# Outputs land in gpurun_out/prof_<round>/ ; tools/parse_round.py <round> turns them into profiles/<round>_*.md + profiles/pmc_traffic.json.
ROUND=${1:-r04}
OUT=gpurun_out/prof_$ROUND
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_driver -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-side > $OUT/stats_driver.log 2>&1 || echo "stats driver failed"
B="python3 bench.py --no-cpu-baseline --no-side --steps 400 --warmup 100"
for W in C2 C3 C4 C5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$W -- $B --workload $W > $OUT/stats_$W.log 2>&1 || echo "stats $W failed"
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_C2_rollout -- $B --mode rollout > $OUT/stats_C2_rollout.log 2>&1 || echo "stats rollout failed"
P="python3 bench.py --no-cpu-baseline --no-side --steps 60 --warmup 10 --launch eager"
for c in FETCH_SIZE WRITE_SIZE; do
  for W in C2 C3 C4 C5; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_${W}_$c -- $P --workload $W > $OUT/pmc_${W}_$c.log 2>&1 || echo "pmc $W $c failed"
  done
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_C2rollout_$c -- $P --mode rollout > $OUT/pmc_C2rollout_$c.log 2>&1 || echo "pmc rollout $c failed"
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_calib_$c -- python3 tools/dbg_launch.py calib > $OUT/pmc_calib_$c.log 2>&1 || echo "pmc calib $c failed"
done
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/sq_step_$i -- $P > $OUT/sq_step_$i.log 2>&1 || echo "sq step $i failed"
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/sq_rollout_$i -- python3 bench.py --no-cpu-baseline --no-side --steps 200 --warmup 20 --mode rollout --reset-prefetch 0 > $OUT/sq_rollout_$i.log 2>&1 || echo "sq rollout $i failed"
done
for c in FETCH_SIZE WRITE_SIZE; do
  for W in C3 C5 X1; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_reset_${W}_$c -- python3 tools/reset_pmc.py $W > $OUT/pmc_reset_${W}_$c.log 2>&1 || echo "pmc reset $W $c failed"
  done
done
# 5. the step with the FUSED LidarInFront observation (in-place step kernel + occupancy bit rows: ngw_step_lean<0, false, ., true, NR>): kernel trace per row format at C2 and
#    with packed rows at C3 / C5, HBM traffic and SQ counters (int16 rows)
for F in int32 int16 packed; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_lidar_$F -- $B --lidar $F > $OUT/stats_lidar_$F.log 2>&1 || echo "stats lidar $F failed"
done
for W in C3 C5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_lidar_${W}_packed -- $B --workload $W --lidar packed > $OUT/stats_lidar_${W}_packed.log 2>&1 || echo "stats lidar $W failed"
done
for c in FETCH_SIZE WRITE_SIZE; do
  for F in int16 packed; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_lidar_${F}_$c -- $P --lidar $F > $OUT/pmc_lidar_${F}_$c.log 2>&1 || echo "pmc lidar $F $c failed"
  done
  for W in C3 C5; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_lidar_${W}_packed_$c -- $P --workload $W --lidar packed > $OUT/pmc_lidar_${W}_packed_$c.log 2>&1 || echo "pmc lidar $W $c failed"
  done
done
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/sq_lidar_$i -- $P --lidar int16 > $OUT/sq_lidar_$i.log 2>&1 || echo "sq lidar $i failed"
done
# 6. in-kernel clock stamps of the step kernel (diagnostics build): the median wave life that bench.py's roofline.floor adds to the empty-kernel launch period
NGW_STAMP_JSON=$OUT/wave_life.json NGW_LIB=$PWD/gym_novel_gridworlds_amd/libngw_hip_stamps.so python3 tools/stamp_timeline.py C2 C3 C4 C5 > $OUT/stamps_step.log 2>&1 || echo "stamps failed"
# 7. untraced rates of the fused lidar step, per workload
mkdir -p gpurun_out/$ROUND; rm -f gpurun_out/$ROUND/lds_layout.log
for W in C2 C3 C5; do
  NGW_LIDAR_SWEEP=0 python3 tools/lidar_rate.py 65536 $W > gpurun_out/$ROUND/lidar_rate_$W.log 2>&1 || echo "lidar_rate $W failed"
done
for W in C2 C3 C5; do echo "== $W" >> gpurun_out/$ROUND/lds_layout.log; NGW_DEBUG_LDS=1 python3 bench.py --no-cpu-baseline --no-side --steps 16 --warmup 2 --workload $W --lidar packed 2>&1 >/dev/null | grep '^\[ngw\]' | sort -u >> gpurun_out/$ROUND/lds_layout.log; done
(git rev-parse --short HEAD 2>/dev/null || cat COMMIT 2>/dev/null || echo unknown) > $OUT/commit.txt
date -u +%Y-%m-%dT%H:%MZ > $OUT/date.txt
find $OUT -name "*.db" -delete 2>/dev/null
du -sh $OUT
echo profile_$ROUND done
