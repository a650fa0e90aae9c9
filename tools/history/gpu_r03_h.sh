#!/bin/bash
# Round-3 visit H: the whole GPU suite after the sampler / lock / queue changes; LoLa unchanged caller (recorded C-ABI trace replayed from C++)
OUT=gpurun_out/r03h
mkdir -p $OUT
timeout 1800 python -m pytest tests -m gpu -q > $OUT/pytest.txt 2>&1; grep -n "passed\|failed\|rror" $OUT/pytest.txt | head -20
timeout 900 python tools/lola_unchanged_caller.py LoLa --reps 20 > $OUT/lola_unchanged_caller.txt 2>&1; cut -c1-300 $OUT/lola_unchanged_caller.txt | tail -10
