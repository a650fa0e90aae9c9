#!/bin/bash
# Round-3 visit I: staged deferral (rotations, MultiplyPlain, SumAllSlots, copies); LoLa unchanged caller with / without deferred submission;
# key-switch workgroup orders under the PMC counters (tools/pmc_ks.sh)
OUT=gpurun_out/r03i
mkdir -p $OUT
timeout 900 python -m pytest tests/test_deferred.py tests/test_lola.py tests/test_gpu_evaluator.py -m gpu -x -q 2>&1 | grep -n "passed\|failed\|rror\|assert" | head -20
timeout 900 python tools/lola_unchanged_caller.py LoLa --reps 20 > $OUT/lola_unchanged_caller.txt 2>&1; cut -c1-330 $OUT/lola_unchanged_caller.txt | tail -14
timeout 1200 bash tools/pmc_ks.sh > $OUT/pmc_ks.txt 2>&1; tail -12 $OUT/pmc_ks.txt | cut -c1-250
