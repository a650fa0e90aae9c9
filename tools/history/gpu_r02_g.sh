#!/bin/bash
# Round-2 visit G: resident-grid transform kernel (k_ntt_rr_stream) - parity, the NTT grid in the three modes, the bench line per mode;
# kernel trace of the bench with the two primes serialised (clean per-kernel durations)
OUT=gpurun_out/r02g
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_evaluator.py -m gpu -x -q -k "long_batch or ntt_roundtrip or behz" > $OUT/pytest.txt 2>&1
tail -3 $OUT/pytest.txt
for m in 0 1 2; do
  CN_NTT_STREAM=$m timeout 300 python tools/ntt_grid.py > $OUT/ntt_grid_$m.txt 2>&1
  echo "== ntt_stream=$m"; grep -E "^C[235] .* (1690|8192) " $OUT/ntt_grid_$m.txt
done
for m in 0 1 2; do
  CN_NTT_STREAM=$m timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-unchanged-caller > $OUT/bench_$m.json 2> $OUT/bench_$m.err
  echo "== bench ntt_stream=$m"; python -c "
import json,sys
d=json.load(open('$OUT/bench_$m.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['ms_per_launch'], d['key_switch']['ms_per_launch'])"
done
export TMPDIR=/tmp
R=$PWD
(cd /tmp && rocprofv3 --kernel-trace --stats -f csv -d $R/$OUT/benchtrace -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-unchanged-caller --serialize > $R/$OUT/bench_traced.json 2> $R/$OUT/bench_traced.err)
KT=$(find $OUT/benchtrace -name "*kernel_trace.csv" | head -1); python tools/summarize_trace.py $KT > $OUT/bench_kernel_trace_summary.txt 2>&1
KS=$(find $OUT/benchtrace -name "*kernel_stats.csv" | head -1); cp $KS $OUT/bench_kernel_stats.csv
head -22 $OUT/bench_kernel_trace_summary.txt | cut -c1-140
find $OUT -name "*kernel_trace.csv" -delete
