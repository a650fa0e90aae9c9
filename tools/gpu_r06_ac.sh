#!/bin/bash
# Round 6, visit AC: the configuration the soak crashed on (seed 1174), repeated
R=$(pwd); O=$R/gpurun_out/r06ac; mkdir -p $O
ulimit -c 0
timeout 600 python tools/soak_lockfree.py --iters 40 --seed 1174 --same-seed --trace > $O/soak1174.txt 2>&1; echo "rc $?" >> $O/soak1174.txt
tail -30 $O/soak1174.txt
