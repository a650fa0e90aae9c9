#!/bin/bash
# visit AM: scalar GEMM / BEHZ units under other scheduling strategies
O=gpurun_out/r03am; mkdir -p $O
for rep in 1 2; do for tag in "" _gmc _gilp _bmc _bilp; do
  lib=$PWD/cryptonets_amd/lib/libcnhip$tag.so
  if [ $rep = 1 ]; then CNHIP_LIB=$lib python -m pytest tests/test_gpu_evaluator.py -q -x -m gpu -k "gemm or multiply_relin or behz" > $O/parity$tag.txt 2>&1; fi
  CNHIP_LIB=$lib python bench.py --steps 20 --warmup 3 --no-unchanged-caller --no-cpu-baseline > $O/bench$tag.json 2>/dev/null
  python -c "
import json; d=json.load(open('$O/bench$tag.json')); print('build [$tag]', open('$O/parity$tag.txt').read().strip().splitlines()[-1][:12], d['value'], d['ms_per_step'], d['verified_against_integer_model'], 'late', d['relinearize_late']['ms_per_step'])"
done; done
