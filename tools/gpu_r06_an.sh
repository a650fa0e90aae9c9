#!/bin/bash
# Round 6, visit AN: the long soaks (other seeds): 15 min of tools/soak_lockfree.py, 2 x 5 min of tools/soak_random_programs.py
R=$(pwd); O=$R/gpurun_out/r06an; mkdir -p $O
ulimit -c 0
timeout 1100 python tools/soak_lockfree.py --seconds 900 --seed 500000 --quiet > $O/soak_lockfree.txt 2>&1; echo "rc $?" >> $O/soak_lockfree.txt; tail -3 $O/soak_lockfree.txt
timeout 500 python tools/soak_random_programs.py --seconds 300 --seed 200000 --params tiny > $O/soak_tiny.txt 2>&1; echo "rc $?" >> $O/soak_tiny.txt; tail -2 $O/soak_tiny.txt
timeout 500 python tools/soak_random_programs.py --seconds 300 --seed 300000 --params c4 --length 200 > $O/soak_c4.txt 2>&1; echo "rc $?" >> $O/soak_c4.txt; tail -2 $O/soak_c4.txt
