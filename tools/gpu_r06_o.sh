#!/bin/bash
# Round 6, visit O: does the alternation of fronts inside the deferred flush engage?  (counters printed when a context goes away, CN_DEFER_TRACE=1)
O=gpurun_out/r06o; mkdir -p $O
CN_DEFER_TRACE=1 python tools/replay_reference_calls.py --trained --threads 16 --steps 8 > $O/replay.txt 2> $O/replay.err
grep "stagger" $O/replay.err | head; python -c "
import json
for ln in open('$O/replay.txt'):
    d = json.loads(ln); print(d['caller'][:40], d['threads'], d['ms_per_batch'], d.get('frac_of_batched'))"
grep -c "flush" $O/replay.err
