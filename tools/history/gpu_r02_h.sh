#!/bin/bash
# Round-2 visit H: k_ntt_rr_stream with LDS twiddles + SGPR first-pass roots: parity, NTT grid per mode, bench per mode
OUT=gpurun_out/r02h
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_evaluator.py -m gpu -x -q -k "long_batch or ntt_roundtrip" > $OUT/pytest.txt 2>&1
tail -3 $OUT/pytest.txt
for m in ${MODES:-0 1}; do
  CN_NTT_STREAM=$m timeout 300 python tools/ntt_grid.py > $OUT/ntt_grid_$m.txt 2>&1
  echo "== ntt_stream=$m"; grep -E "^C[235] .* (1690|8192) " $OUT/ntt_grid_$m.txt
done
for m in ${MODES:-0 1}; do
  CN_NTT_STREAM=$m timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-unchanged-caller > $OUT/bench_$m.json 2> $OUT/bench_$m.err
  echo "== bench ntt_stream=$m"; python -c "
import json,sys
d=json.load(open('$OUT/bench_$m.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['ms_per_launch'], d['key_switch']['ms_per_launch'])"
done
