#!/bin/bash
# Performance matrix of a round: tools/perf_matrix.sh <round, e.g. r03> - the default bench line of every workload (with its side
# measurements), the driver-style short run, the 2-rank rehearsal of the multi-GPU leg on one GPU, reset launch times and the
# in-kernel timelines (the diagnostics build is REBUILT here first, so it can never be stale).  tools/write_perf_matrix.py <round>
# turns the output into profiles/<round>_perf_matrix.md.
ROUND=${1:-r04}
OUT=gpurun_out/matrix_$ROUND; rm -rf $OUT; mkdir -p $OUT
make -C gym_novel_gridworlds_amd/csrc stamps > $OUT/stamps_build.log 2>&1 || { echo "stamps build failed"; tail -5 $OUT/stamps_build.log; }
for W in C2 C3 C4 C5; do
  timeout -k 10 300 python bench.py --workload $W > $OUT/bench_$W.log 2>&1 || echo "bench $W failed"
  grep '^{' $OUT/bench_$W.log | tail -1 > $OUT/bench_$W.json
done
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_C2_driver_style.log 2>&1; grep '^{' $OUT/bench_C2_driver_style.log | tail -1 > $OUT/bench_C2_driver_style.json
for W in C4 C5; do
  timeout -k 10 300 python bench.py --gpus 2 --dist-backend gloo --single-device --workload $W --no-cpu-baseline --steps 400 > $OUT/bench_2rank_$W.log 2>&1; grep '^{' $OUT/bench_2rank_$W.log | tail -1 > $OUT/bench_2rank_$W.json
done
for W in X1 X2 X3; do   # wrapper predicates (FireWall, FenceRestriction, Crate): high episode churn, adaptive prepared episodes
  timeout -k 10 300 python bench.py --workload $W --no-cpu-baseline > $OUT/bench_$W.log 2>&1; grep '^{' $OUT/bench_$W.log | tail -1 > $OUT/bench_$W.json
done
NGW_HOST_DELTA=0 timeout -k 10 200 python tools/api_mode_rate.py > $OUT/api_full_copy.log 2>&1
timeout -k 10 200 python tools/api_mode_rate.py > $OUT/api_delta.log 2>&1
timeout -k 10 100 python tools/short_run.py 20 > $OUT/short_run.log 2>&1
timeout -k 10 300 python tools/reset_time.py C2 C3 C4 C5 X1 X2 X3 > $OUT/reset_time.log 2>&1
NGW_FAST_RESET=0 timeout -k 10 300 python tools/reset_time.py > $OUT/reset_time_general.log 2>&1
timeout -k 10 100 python tools/adapter_latency.py > $OUT/adapter.log 2>&1
timeout -k 10 200 python tools/api_latency.py > $OUT/api.log 2>&1
timeout -k 10 300 python tools/lidar_rate.py > $OUT/lidar.log 2>&1
ROUND=matrix_$ROUND bash tools/ab_stage.sh > $OUT/ab_stage.log 2>&1
if [ -f gym_novel_gridworlds_amd/libngw_hip_stamps.so ]; then
  NGW_LIB=$PWD/gym_novel_gridworlds_amd/libngw_hip_stamps.so timeout -k 10 300 python tools/stamp_timeline.py C2 C3 C4 C5 > $OUT/stamps.log 2>&1
  NGW_LIB=$PWD/gym_novel_gridworlds_amd/libngw_hip_stamps.so NGW_NOSTAGE=0 timeout -k 10 300 python tools/stamp_timeline.py C2 > $OUT/stamps_staged.log 2>&1
  NGW_LIB=$PWD/gym_novel_gridworlds_amd/libngw_hip_stamps.so timeout -k 10 300 python tools/stamp_reset.py C3 C5 X1 > $OUT/stamps_reset.log 2>&1
fi
echo matrix done
