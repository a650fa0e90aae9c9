#!/bin/bash
# Round 6, visit AH: how far ahead of the device do the callers of the unchanged sequence run?  (REPLAY_HOST_TIMES: when rp.run returned for every batch, no wait for the device)
R=$(pwd); O=$R/gpurun_out/r06ah; mkdir -p $O
REPLAY_HOST_TIMES=1 python tools/replay_reference_calls.py --trained --threads 16 --literal-threads 16 --steps 24 > $O/run.txt 2> $O/host_times.txt
cat $O/host_times.txt | cut -c1-400; cut -c1-200 $O/run.txt
