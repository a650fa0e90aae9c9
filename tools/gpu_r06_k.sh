#!/bin/bash
# Round 6, visit K: staggered plaintext-prime channels inside the deferred flush (CN_DEFER_STAGGER=1, default) against lock-step flushes: the unchanged caller; parity of the deferred suite
O=gpurun_out/r06k; mkdir -p $O
timeout 900 python -m pytest tests/test_deferred.py tests/test_cryptonets_mnist.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
for rep in 1 2 3; do for st in 0 1; do
  CN_DEFER_STAGGER=$st python tools/replay_reference_calls.py --trained --threads 16 --literal-threads 16,256 --steps 5 > $O/replay.txt 2> $O/replay.err
  python -c "
import json
for ln in open('$O/replay.txt'):
    d = json.loads(ln); print('stagger $st rep $rep:', d['caller'][:40], d['threads'], d['ms_per_batch'], d.get('frac_of_batched'), d.get('words_identical'))"
done; done
