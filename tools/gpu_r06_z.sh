#!/bin/bash
# Round 6, visit Z: queued MultiplyPlain calls that share their ciphertext take the broadcast form at flush (the per-row DotProduct of the LoLa dense layers): parity, then the unchanged LoLa caller
O=gpurun_out/r06z; mkdir -p $O
timeout 1200 python -m pytest tests/test_lola.py tests/test_deferred.py tests/test_layers.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
timeout 600 python tools/lola_unchanged_caller.py LoLa --reps 20 > $O/lola.txt 2> $O/lola.err; python -c "
import json
for ln in open('$O/lola.txt'):
    d = json.loads(ln); print(d['pattern'][:70], '|', d['host'][:40], d['ms_per_image'], d['logits_exact'], d.get('launches_per_prime'))"
