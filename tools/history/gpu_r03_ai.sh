#!/bin/bash
# visit AI: per-digit vs per-source-limb workgroups of the two-launch key switch for the 10-13-ciphertext steps of the dense layers
for d in 0 10 20 32; do
echo "CN_KS_DIGIT_MAX=$d"
CN_KS_DIGIT_MAX=$d python tools/chain_concurrency_probe.py LoLa 2>&1 | grep "contexts \[0\] \|contexts \[0, 1, 2, 3\]"
CN_KS_DIGIT_MAX=$d python bench.py --workload lola --steps 20 --warmup 2 --no-unchanged-caller 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  bench lola', d['value'], d['ms_per_step'], d['verified_against_integer_model'])"
done
