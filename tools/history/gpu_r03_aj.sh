#!/bin/bash
# visit AJ2: transform units with -amdgpu-sched-strategy=max-ilp, key-switch units default / max-memory-clause
O=gpurun_out/r03aj; mkdir -p $O
for tag in "" _rrilp _rrilp_ksmc "" _rrilp _rrilp_ksmc; do
  lib=$PWD/cryptonets_amd/lib/libcnhip$tag.so
  if [ ! -f $O/parity$tag.txt ]; then CNHIP_LIB=$lib python -m pytest tests/test_gpu_evaluator.py -q -x -m gpu -k "key_switch or multiply_relin or ntt_roundtrip or squaring or multiply_plain" > $O/parity$tag.txt 2>&1; grep -E "passed|failed" $O/parity$tag.txt | tail -1; fi
  CNHIP_LIB=$lib python bench.py --steps 20 --warmup 3 --no-unchanged-caller --no-cpu-baseline > $O/bench$tag.json 2>/dev/null
  python -c "
import json; d=json.load(open('$O/bench$tag.json')); print('build [$tag]', d['value'], d['ms_per_step'], d['verified_against_integer_model'], 'ntt frac', d['roofline']['frac'], 'ks ms', d['key_switch']['ms_per_launch'], 'late', d['relinearize_late']['ms_per_step'])"
done
