#!/bin/bash
# Round 5, visit G: the forward FP64 transform of 45-49-bit moduli recentres once instead of three times - whole GPU suite (every word must stay), probe, CIFAR line, default line
O=gpurun_out/r05g; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
timeout 300 python tools/ks14_probe.py 5488 ks_pair14=1 2>&1 | tee $O/ks14_probe.txt
python bench.py --workload cifar --steps 3 --warmup 2 > $O/cifar.json 2> $O/cifar.err; tail -1 $O/cifar.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_image'], d['verified_against_integer_model']); print(json.dumps(d.get('key_switch'))[:1800])"
python bench.py --steps 10 --warmup 3 --no-single-image > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['verified_against_integer_model'], d['roofline']['frac'], d['key_switch']['ms_per_launch'], d['square']['ms_per_chain'], d['unchanged_caller']['frac_of_batched'])"
