#!/bin/bash
# round 5, visit ac: where the unchanged LoLa caller's extra 1.2 ms per image goes - flush host times (CN_DEFER_TRACE=2) of the replayed call sequence
O=gpurun_out/r05ac; mkdir -p $O
python tools/lola_unchanged_caller.py LoLa --reps 20 > $O/lola_unchanged.txt 2> $O/err1.txt; cat $O/lola_unchanged.txt | cut -c1-260
CN_DEFER_TRACE=2 python tools/lola_unchanged_caller.py LoLa --reps 3 > $O/trace_out.txt 2> $O/trace_err.txt
grep -c "flush" $O/trace_err.txt; grep "flush" $O/trace_err.txt | tail -120 > $O/flush_tail.txt; tail -60 $O/flush_tail.txt | cut -c1-200
