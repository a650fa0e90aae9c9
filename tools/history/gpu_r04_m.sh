#!/bin/bash
# round 4, visit m: matrix-core scalar GEMM with two steps of input prefetch (ring of three register sets): words + kernel time
OUT=gpurun_out/r04m
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_evaluator.py tests/test_cryptonets_mnist.py tests/test_deferred.py tests/test_lola.py -m gpu -x -q -k "gemm or end_to_end or unchanged or lola or deferred" > $OUT/pytest.txt 2>&1
tail -3 $OUT/pytest.txt
export TMPDIR=/tmp
R=$PWD
(cd /tmp && rocprofv3 --kernel-trace --stats -f csv -d $R/$OUT/prof -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-unchanged-caller --no-single-image --no-relinearize-late --serialize > $R/$OUT/prof_bench.json 2> $R/$OUT/prof.err)
KT=$(find $OUT/prof -name "*kernel_trace.csv" | head -1)
python tools/summarize_trace.py $KT > $OUT/trace_summary.txt 2>&1
rm -rf $OUT/prof
grep -E "gemm|kernel " $OUT/trace_summary.txt
for i in 1 2; do python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-unchanged-caller --no-single-image --no-relinearize-late 2>> $OUT/bench.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['verified_against_integer_model'])"; done
