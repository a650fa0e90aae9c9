#!/bin/bash
# visit AK: FP64 key-switch units built with -amdgpu-sched-strategy=max-memory-clause (default build now) against the compiler's default strategy
# (libcnhip_nosched.so): parity, CryptoNets batch, LoLa-MNIST, LoLa-CIFAR
O=gpurun_out/r03ak; mkdir -p $O
python -m pytest tests/test_gpu_evaluator.py tests/test_lola.py tests/test_lola_cifar.py tests/test_deferred.py -q -x -m gpu > $O/parity.txt 2>&1; grep -E "passed|failed" $O/parity.txt | tail -1
for rep in 1 2; do for tag in "" _nosched; do
  lib=$PWD/cryptonets_amd/lib/libcnhip$tag.so
  CNHIP_LIB=$lib python bench.py --steps 20 --warmup 3 --no-unchanged-caller --no-cpu-baseline > $O/bench$tag.json 2>/dev/null
  python -c "
import json; d=json.load(open('$O/bench$tag.json')); print('build [$tag]', d['value'], d['ms_per_step'], d['verified_against_integer_model'], 'ntt frac', d['roofline']['frac'], 'ks ms', d['key_switch']['ms_per_launch'], 'frac valu', d['key_switch'].get('frac_valu_in_situ'), 'late', d['relinearize_late']['ms_per_step'])"
  CNHIP_LIB=$lib python bench.py --workload lola --steps 20 --warmup 2 --no-unchanged-caller 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('   lola', d['value'], d['ms_per_step'], d['verified_against_integer_model'])"
done; done
for tag in "" _nosched; do CNHIP_LIB=$PWD/cryptonets_amd/lib/libcnhip$tag.so python bench.py --workload cifar --steps 2 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('build [$tag] cifar', d['value'], d['ms_per_step'], d['verified_against_integer_model'])"; done
