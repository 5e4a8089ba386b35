#!/bin/bash
# Where the fused LidarInFront epilogue's time goes: in-kernel stamps (prebuilt diagnostics library) + the rate table.
OUT=gpurun_out/${ROUND:-r04}; mkdir -p $OUT
for dt in int16 packed int32; do
  NGW_LIDAR=1 NGW_LIDAR_DTYPE=$dt NGW_LIB=$PWD/gym_novel_gridworlds_amd/libngw_hip_stamps.so timeout -k 10 200 python tools/stamp_timeline.py C2 > $OUT/stamp_lidar_$dt.log 2>&1
  grep -v amdgpu.ids $OUT/stamp_lidar_$dt.log
done
timeout -k 10 300 python tools/lidar_rate.py > $OUT/lidar_rate.log 2>&1; grep -v amdgpu.ids $OUT/lidar_rate.log
