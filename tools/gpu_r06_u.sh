#!/bin/bash
# Round 6, visit U: queued scalar products through INDEX tables (256-byte offsets from the lowest address, CN_DEFER_REL=1, default) against address tables; + gather lists in pairs on top: parity, then the unchanged caller
O=gpurun_out/r06u; mkdir -p $O
timeout 1200 python -m pytest tests/test_deferred.py tests/test_cryptonets_mnist.py tests/test_layers.py tests/test_lola.py tests/test_basic_operations.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
CN_DEFER_PAIR=1 timeout 900 python -m pytest tests/test_deferred.py tests/test_cryptonets_mnist.py -m gpu -x -q > $O/pytest_pair.txt 2>&1; tail -1 $O/pytest_pair.txt
for rep in 1 2 3; do for mode in "0 0" "1 0" "1 1"; do
  set -- $mode
  CN_DEFER_REL=$1 CN_DEFER_PAIR=$2 python tools/replay_reference_calls.py --trained --threads 16 --literal-threads 16,256 --steps 5 > $O/replay.txt 2> $O/replay.err
  python -c "
import json
for ln in open('$O/replay.txt'):
    d = json.loads(ln)
    if d['threads'] != 1: print('rel $1 pair $2 rep $rep:', d['caller'][:40], d['threads'], d['ms_per_batch'], d.get('frac_of_batched'), d.get('words_identical'))"
done; done
