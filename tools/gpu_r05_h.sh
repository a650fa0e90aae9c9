#!/bin/bash
# Round 5, visit H: the unchanged CryptoNets caller at 16 and 256 caller threads: flush host times (CN_DEFER_TRACE=2), launches, fraction of batched
O=gpurun_out/r05h; mkdir -p $O
nproc > $O/nproc.txt; cat /sys/fs/cgroup/cpu.max >> $O/nproc.txt 2>/dev/null
python tools/replay_reference_calls.py --trained --threads 16,256 --literal-threads 16,64,256 --steps 5 > $O/replay.txt 2> $O/replay.err; cat $O/replay.txt | cut -c1-330
CN_DEFER_TRACE=2 python tools/replay_reference_calls.py --trained --threads 16 --literal-threads 16 --steps 2 > $O/trace16.txt 2> $O/trace16.err
CN_DEFER_TRACE=2 python tools/replay_reference_calls.py --trained --threads 256 --literal-threads 256 --steps 2 > $O/trace256.txt 2> $O/trace256.err
tail -60 $O/trace16.err | cut -c1-200
echo ===== 256
tail -60 $O/trace256.err | cut -c1-200
