#!/bin/bash
# The whole -m gpu suite once per A/B switch of INTEGRATION.md (through gpurun, from the repo root; ~1.5 minutes per switch).
# A test failure is recorded and the next switch runs; a run that is killed or times out ends the script (no further GPU step after a kill).
OUT=gpurun_out/${ROUND:-r05}/switches; mkdir -p $OUT
SWITCHES=${*:-"NGW_LIDAR_BOARDS=0 NGW_HOST_WRITE_THROUGH=0 NGW_SOLO=0 NGW_API_SLICES=2 NGW_LDS_ALIAS=0 NGW_FAST_RESET=0 NGW_FAST_RESET=2 NGW_NOSTAGE=0 NGW_HOST_DELTA=0 NGW_ADAPT_PREFETCH=0"}
: > $OUT/summary.txt
for sw in $SWITCHES; do
  env $sw timeout -k 10 400 python -m pytest tests -q -m gpu -p no:cacheprovider > $OUT/$sw.log 2>&1
  rc=$?
  echo "$sw: rc=$rc: $(tail -1 $OUT/$sw.log)" | tee -a $OUT/summary.txt
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed or timed out under $sw: stopping" | tee -a $OUT/summary.txt; exit 1; fi
done
