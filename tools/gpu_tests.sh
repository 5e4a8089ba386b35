#!/bin/bash
# Whole -m gpu suite on the box; the log is kept under gpurun_out/r03.
OUT=gpurun_out/${ROUND:-r04}; mkdir -p $OUT
timeout -k 10 1000 python -m pytest tests -x -q -m gpu "$@" > $OUT/gputest.log 2>&1; rc=$?; tail -15 $OUT/gputest.log; exit $rc
