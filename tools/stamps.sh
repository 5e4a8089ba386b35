#!/bin/bash
# In-kernel timelines (diagnostics build, always rebuilt here so it can never be stale): tools/stamps.sh <reset workloads> -- <step workloads>
OUT=gpurun_out/${ROUND:-r04}; mkdir -p $OUT
make -C gym_novel_gridworlds_amd/csrc stamps > $OUT/stamps_build.log 2>&1 || { tail -20 $OUT/stamps_build.log; exit 1; }
export NGW_LIB=$PWD/gym_novel_gridworlds_amd/libngw_hip_stamps.so
R=(); T=(); cur=R
for x in "$@"; do if [ "$x" = "--" ]; then cur=T; elif [ $cur = R ]; then R+=($x); else T+=($x); fi; done
[ ${#R[@]} -gt 0 ] && { timeout -k 10 300 python tools/stamp_reset.py ${R[@]} > $OUT/stamp_reset.log 2>&1; grep -v amdgpu.ids $OUT/stamp_reset.log; }
[ ${#T[@]} -gt 0 ] && { timeout -k 10 300 python tools/stamp_timeline.py ${T[@]} > $OUT/stamp_step.log 2>&1; grep -v amdgpu.ids $OUT/stamp_step.log; }
exit 0
