#!/bin/bash
# Round 5, visit V: the unchanged LoLa-MNIST caller - flush host times per queue level (CN_DEFER_TRACE=2) of one image
O=gpurun_out/r05v; mkdir -p $O
CN_DEFER_TRACE=2 python tools/lola_unchanged_caller.py LoLa --reps 3 > $O/lola.txt 2> $O/lola.err
grep -c "flush of" $O/lola.err
# the last image of the deferred replay with one thread per prime: flushes of ONE context
ctx=$(grep "flush of" $O/lola.err | tail -1 | awk '{print $2}')
grep "$ctx" $O/lola.err | tail -120 | grep "flush of" | awk '{n++; s+=$(NF-4)} END {print n, "flushes of the last lines,", s, "us"}'
grep "$ctx" $O/lola.err | tail -70 | cut -c1-150
tail -6 $O/lola.txt | cut -c1-300
