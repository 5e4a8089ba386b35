#!/bin/bash
# Round 3, first GPU call: parity of the new subset passes + reset timings.
OUT=gpurun_out/${ROUND:-r04}; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "reset or additem or distribution or wall_ring or prepared or staggered or placement" > $OUT/reset_tests.log 2>&1; tail -5 $OUT/reset_tests.log
for F in 1 0; do NGW_FAST_RESET=$F timeout -k 10 200 python tools/reset_time.py C3 C5 X1 > $OUT/reset_time_$F.log 2>&1; grep -v amdgpu.ids $OUT/reset_time_$F.log | tail -4; done
