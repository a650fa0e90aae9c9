#!/bin/bash
# round 4, visit r: what is the one-time ~55 ms in the first / second batch behind new input handles of the unchanged caller?  flush host times next to the per-batch times
OUT=gpurun_out/r04r
mkdir -p $OUT
CN_DEFER_TRACE=2 python tools/replay_reference_calls.py --trained --threads 1 --literal-threads 16,16,16 --steps 3 --warmup 1 --per-step 2> $OUT/trace.txt > /dev/null
grep -E "step|flush of" $OUT/trace.txt | awk '/step/ {print; next} { if ($(NF-4)+0 > 2000) print "      SLOW FLUSH:", $0 }' | head -60
