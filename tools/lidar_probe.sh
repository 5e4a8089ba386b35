#!/bin/bash
# Where the fused LidarInFront epilogue's time goes: in-kernel stamps + the rate table.  The diagnostics library is REBUILT here from the
# sources of the snapshot, so a timeline can never describe another build than the one that was sent (round 4 once parsed the timelines of a dropped experiment into profiles/).
OUT=gpurun_out/${ROUND:-r04}; mkdir -p $OUT
make -C gym_novel_gridworlds_amd/csrc stamps > $OUT/stamps_build.log 2>&1 || { echo "stamps build failed"; tail -5 $OUT/stamps_build.log; exit 1; }
for dt in int16 packed int32; do
  NGW_LIDAR=1 NGW_LIDAR_DTYPE=$dt NGW_LIB=$PWD/gym_novel_gridworlds_amd/libngw_hip_stamps.so timeout -k 10 200 python tools/stamp_timeline.py C2 > $OUT/stamp_lidar_$dt.log 2>&1
  grep -v amdgpu.ids $OUT/stamp_lidar_$dt.log
done
timeout -k 10 300 python tools/lidar_rate.py > $OUT/lidar_rate.log 2>&1; grep -v amdgpu.ids $OUT/lidar_rate.log
