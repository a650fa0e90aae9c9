#!/bin/bash
# Round-2 visit M: GEMM epilogues skip the bias scaling of zero coefficients: parity, bench, GEMM kernel times
OUT=gpurun_out/r02m
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_evaluator.py tests/test_cryptonets_mnist.py tests/test_deferred.py -m gpu -x -q -k "gemm or cryptonets or defer or dot" > $OUT/pytest.txt 2>&1
tail -3 $OUT/pytest.txt
export TMPDIR=/tmp
R=$PWD
for t in 1 2; do
  timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-unchanged-caller > $OUT/bench_$t.json 2> $OUT/bench_$t.err
  echo "== bench: $(python -c "import json; d=json.load(open('$OUT/bench_$t.json')); print(d['value'], d['ms_per_step'], d['verified_against_integer_model'])")"
done
(cd /tmp && rocprofv3 --kernel-trace --stats -f csv -d $R/$OUT/prof -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-unchanged-caller --serialize > /dev/null 2> $R/$OUT/prof.err)
KT=$(find $OUT/prof -name "*kernel_trace.csv" | head -1); python tools/summarize_trace.py $KT > $OUT/trace.txt 2>&1; find $OUT/prof -name "*kernel_trace.csv" -delete
grep -E "gemm" $OUT/trace.txt | cut -c1-120
