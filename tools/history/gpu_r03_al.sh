#!/bin/bash
# visit AL: more scheduling variants of the <= 44-bit key-switch unit against the kept build (max-memory-clause)
O=gpurun_out/r03al; mkdir -p $O
for rep in 1 2; do for tag in "" _itminreg _itmaxocc _mcbias0 _mcbias50 _mctrack; do
  lib=$PWD/cryptonets_amd/lib/libcnhip$tag.so
  if [ $rep = 1 ]; then CNHIP_LIB=$lib python -m pytest tests/test_gpu_evaluator.py -q -x -m gpu -k "key_switch or multiply_relin" > $O/parity$tag.txt 2>&1; fi
  CNHIP_LIB=$lib python bench.py --steps 20 --warmup 3 --no-unchanged-caller --no-cpu-baseline --no-relinearize-late > $O/bench$tag.json 2>/dev/null
  python -c "
import json; d=json.load(open('$O/bench$tag.json')); print('build [$tag]', open('$O/parity$tag.txt').read().strip().splitlines()[-1][:12], d['value'], d['ms_per_step'], d['verified_against_integer_model'], 'ks ms', d['key_switch']['ms_per_launch'])"
done; done
