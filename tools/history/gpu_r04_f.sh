#!/bin/bash
# round 4, visit f: scalar GEMM of the convolution with two gather lists in flight per thread (PF = 8): words + kernel time
OUT=gpurun_out/r04f
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_evaluator.py tests/test_cryptonets_mnist.py tests/test_deferred.py tests/test_layers.py -m gpu -x -q -k "gemm or end_to_end or unchanged or literal or layers or Pool" > $OUT/pytest.txt 2>&1
tail -3 $OUT/pytest.txt
export TMPDIR=/tmp
R=$PWD
(cd /tmp && rocprofv3 --kernel-trace --stats -f csv -d $R/$OUT/prof -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-unchanged-caller --no-single-image --no-relinearize-late --serialize > $R/$OUT/prof_bench.json 2> $R/$OUT/prof.err)
KT=$(find $OUT/prof -name "*kernel_trace.csv" | head -1)
python tools/summarize_trace.py $KT > $OUT/trace_summary.txt 2>&1
rm -rf $OUT/prof
grep -E "gemm|kernel " $OUT/trace_summary.txt
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-single-image --no-relinearize-late > $OUT/bench.json 2>> $OUT/bench.err
python -c "import json; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); u=d['unchanged_caller']; print(d['ms_per_step'], d['verified_against_integer_model'], 'unchanged', u['ms_per_step'], u['frac_of_batched'], u['verified_against_integer_model'])"
