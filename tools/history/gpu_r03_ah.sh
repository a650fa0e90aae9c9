#!/bin/bash
# visit AH: up to how many (ciphertext, limb) blocks should a key switch run as two launches, now that four chains run side by side?
for w in 160 64 48 32 20; do
echo "CN_KS_WIDE_MAX=$w"
CN_KS_WIDE_MAX=$w python tools/chain_concurrency_probe.py LoLa 2>&1 | grep "contexts \[0\] \|contexts \[0, 1, 2, 3\]"
CN_KS_WIDE_MAX=$w python bench.py --workload lola --no-unchanged-caller 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  bench lola', d['value'], d['ms_per_step'], d['verified_against_integer_model'])"
done
