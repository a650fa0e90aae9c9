#!/bin/bash
# Round-2 visit L: block-sparse tiled convolution (k_scalar_gemm_f64<20, .., SKIP>): parity, bench A/B BENCH_CONV_TILE=1|2 with kernel time
OUT=gpurun_out/r02l
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_evaluator.py tests/test_cryptonets_mnist.py -m gpu -x -q -k "gemm or cryptonets" > $OUT/pytest.txt 2>&1
tail -3 $OUT/pytest.txt
export TMPDIR=/tmp
R=$PWD
for t in 1 2 1 2; do
  BENCH_CONV_TILE=$t timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-unchanged-caller > $OUT/bench_$t.json 2> $OUT/bench_$t.err
  echo "== conv tile $t: $(python -c "import json; d=json.load(open('$OUT/bench_$t.json')); print(d['value'], d['ms_per_step'], d['verified_against_integer_model'])")"
done
for t in 1 2; do
  (cd /tmp && BENCH_CONV_TILE=$t rocprofv3 --kernel-trace --stats -f csv -d $R/$OUT/prof$t -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-unchanged-caller --serialize > /dev/null 2> $R/$OUT/prof$t.err)
  KT=$(find $OUT/prof$t -name "*kernel_trace.csv" | head -1); python tools/summarize_trace.py $KT > $OUT/trace$t.txt 2>&1; find $OUT/prof$t -name "*kernel_trace.csv" -delete
  echo "== trace tile $t"; grep -E "gemm" $OUT/trace$t.txt | cut -c1-120
done
