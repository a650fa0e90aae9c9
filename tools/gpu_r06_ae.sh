#!/bin/bash
# Round 6, visit AE: soak of the deferred queue's hazard logic (tools/soak_random_programs.py), tiny ring and the LoLa ring
R=$(pwd); O=$R/gpurun_out/r06ae; mkdir -p $O
ulimit -c 0
timeout 500 python tools/soak_random_programs.py --seconds 240 --seed 100 --params tiny > $O/soak_tiny.txt 2>&1; echo "rc $?" >> $O/soak_tiny.txt; tail -4 $O/soak_tiny.txt
timeout 500 python tools/soak_random_programs.py --seconds 180 --seed 5000 --params c4 --length 150 > $O/soak_c4.txt 2>&1; echo "rc $?" >> $O/soak_c4.txt; tail -4 $O/soak_c4.txt
timeout 300 python -m pytest tests/test_deferred.py -q -m gpu -k "soak or lockfree or zero_enc" 2>&1 | tail -3
