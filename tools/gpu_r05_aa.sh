#!/bin/bash
# round 5, visit aa: k_keyswitch_rr with the source words of the next limb requested a limb ahead: words, batch time A/B against -DKS_SRC_AHEAD=0
O=gpurun_out/r05aa; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_evaluator.py tests/test_cryptonets_mnist.py tests/test_deferred.py -m gpu -x -q -k "key_switch or relin or rotat or multiply or squar or galois or sum_slots or end_to_end or deferred" > $O/pytest.txt 2>&1
tail -2 $O/pytest.txt
L=$PWD/cryptonets_amd/lib
for v in noahead "" noahead "" noahead ""; do
  CNHIP_LIB=$L/libcnhip${v:+_$v}.so python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-unchanged-caller --no-single-image --no-relinearize-late 2>> $O/bench.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${v:-ahead}', d['ms_per_step'], d['value'], d['verified_against_integer_model'], d['key_switch']['ms_per_launch'], d['key_switch'].get('frac'))"
done | tee $O/bench_ab.txt
