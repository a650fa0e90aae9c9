#!/bin/bash
# Round-2 closing visit (run as r02k, again as r02p after the last kernel changes): full GPU suite, smoke, the default bench line, the LoLa / CIFAR bench lines, a serialised kernel trace of the bench
OUT=gpurun_out/r02p
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.txt 2>&1
tail -3 $OUT/pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-120
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
cut -c1-200 $OUT/bench.json; tail -1 $OUT/bench.err | cut -c1-200
timeout 600 python bench.py --workload lola --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_lola.json 2> $OUT/bench_lola.err; cut -c1-300 $OUT/bench_lola.json
timeout 600 python bench.py --workload cifar --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_cifar.json 2> $OUT/bench_cifar.err; cut -c1-300 $OUT/bench_cifar.json
export TMPDIR=/tmp
R=$PWD
(cd /tmp && rocprofv3 --kernel-trace --stats -f csv -d $R/$OUT/benchtrace -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-unchanged-caller --serialize > $R/$OUT/bench_traced.json 2> $R/$OUT/bench_traced.err)
KT=$(find $OUT/benchtrace -name "*kernel_trace.csv" | head -1); python tools/summarize_trace.py $KT > $OUT/bench_kernel_trace_summary.txt 2>&1
KS=$(find $OUT/benchtrace -name "*kernel_stats.csv" | head -1); cp $KS $OUT/bench_kernel_stats.csv
head -20 $OUT/bench_kernel_trace_summary.txt | cut -c1-140
find $OUT -name "*kernel_trace.csv" -delete
