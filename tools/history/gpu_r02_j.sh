#!/bin/bash
# Round-2 visit J: key prefetch in the fused key switch (private registers): parity, bench A/B
OUT=gpurun_out/r02j
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_evaluator.py -m gpu -x -q -k "key_switch or relin or rotat or galois or behz" > $OUT/pytest.txt 2>&1
tail -3 $OUT/pytest.txt
for m in 0 1 0 1; do
  CN_KS_PREFETCH=$m timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-unchanged-caller > $OUT/bench_$m.json 2> $OUT/bench_$m.err
  echo "== bench ks_prefetch=$m"; python -c "
import json,sys
d=json.load(open('$OUT/bench_$m.json')); print(d['value'], d['ms_per_step'], d['key_switch']['ms_per_launch'], d['key_switch'].get('frac_valu_in_situ'))"
done
