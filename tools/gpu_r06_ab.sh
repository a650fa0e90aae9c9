#!/bin/bash
# Round 6, visit AB: soak of the per-ciphertext caller on the deferred queue (tools/soak_lockfree.py), 8 minutes of random configurations
R=$(pwd); O=$R/gpurun_out/r06ab; mkdir -p $O
timeout 900 python tools/soak_lockfree.py --seconds 480 --seed 1000 > $O/soak.txt 2>&1; echo "rc $?" >> $O/soak.txt
tail -5 $O/soak.txt; grep -c "ok:" $O/soak.txt
