#!/bin/bash
# Round 6, visit W: the alternation of fronts inside the deferred flush over LONG windows (20 batches: the trailing context's 2.5 ms at the end of a window weigh a fourth of what they do in 5)
O=gpurun_out/r06w; mkdir -p $O
for rep in 1 2 3; do for st in 0 1; do
  CN_DEFER_STAGGER=$st python tools/replay_reference_calls.py --trained --threads 16 --literal-threads 16 --steps 20 > $O/replay.txt 2> $O/replay.err
  python -c "
import json
for ln in open('$O/replay.txt'):
    d = json.loads(ln)
    if d['threads'] != 1: print('stagger $st rep $rep:', d['caller'][:40], d['threads'], d['ms_per_batch'], d.get('frac_of_batched'), d.get('words_identical'))"
done; done
