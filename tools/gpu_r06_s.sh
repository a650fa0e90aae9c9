#!/bin/bash
# Round 6, visit S: gather lists of queued scalar products merged in pairs at flush (CN_DEFER_PAIR=1, default) against one list per window: parity, then the unchanged caller, alternating
O=gpurun_out/r06s; mkdir -p $O
timeout 900 python -m pytest tests/test_deferred.py tests/test_cryptonets_mnist.py tests/test_layers.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
for rep in 1 2 3; do for pr in 0 1; do
  CN_DEFER_PAIR=$pr python tools/replay_reference_calls.py --trained --threads 16 --literal-threads 16 --steps 5 > $O/replay.txt 2> $O/replay.err
  python -c "
import json
for ln in open('$O/replay.txt'):
    d = json.loads(ln)
    if d['threads'] == 16: print('pair $pr rep $rep:', d['caller'][:40], d['threads'], d['ms_per_batch'], d.get('frac_of_batched'), d.get('words_identical'))"
done; done
