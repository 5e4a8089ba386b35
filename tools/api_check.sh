#!/bin/bash
# Host-API (PCIe-inclusive) rate with and without the delta refresh + the delta parity tests
OUT=gpurun_out/${ROUND:-r04}; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "host_step or big_batch or one_wavefront" > $OUT/api_tests.log 2>&1; tail -4 $OUT/api_tests.log
for D in 1 0; do NGW_HOST_DELTA=$D timeout -k 10 200 python tools/api_mode_rate.py > $OUT/api_rate_$D.log 2>&1; echo "NGW_HOST_DELTA=$D: $(grep -v amdgpu.ids $OUT/api_rate_$D.log | tail -3)"; done
