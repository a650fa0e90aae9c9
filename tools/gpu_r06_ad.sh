#!/bin/bash
# Round 6, visit AD: the soak again from its first seed (the crash needed ~170 networks of history), plain and under rocgdb
R=$(pwd); O=$R/gpurun_out/r06ad; mkdir -p $O
ulimit -c 0
timeout 400 python tools/soak_lockfree.py --seconds 200 --seed 1000 --trace > $O/soak_plain.txt 2>&1; echo "rc $?" >> $O/soak_plain.txt
tail -40 $O/soak_plain.txt | cut -c1-300
timeout 600 /opt/rocm/bin/rocgdb -batch -ex "handle SIGSEGV stop print" -ex run -ex bt -ex "info threads" -ex "thread apply all bt 12" --args python tools/soak_lockfree.py --seconds 240 --seed 1000 --quiet > $O/soak_gdb.txt 2>&1
tail -150 $O/soak_gdb.txt | cut -c1-300
