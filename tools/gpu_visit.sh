#!/bin/bash
# One GPU visit (run through gpurun): GPU test-suite, bench line, rocprofv3 kernel trace of the same command, latency tools.
# Everything lands under gpurun_out/visit/; copy what should be judged into profiles/.
OUT=gpurun_out/visit
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.txt 2>&1
tail -4 $OUT/pytest.txt
python bench.py --steps 5 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err
cut -c1-230 $OUT/bench.json
export TMPDIR=/tmp
R=$PWD
(cd /tmp && rocprofv3 --kernel-trace --stats -f csv -d $R/$OUT/prof -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --serialize > $R/$OUT/prof_bench.json 2> $R/$OUT/prof.err)
KT=$(find $OUT/prof -name "*kernel_trace.csv" | head -1)
python tools/summarize_trace.py $KT > $OUT/trace_summary.txt 2>&1
find $OUT/prof -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/prof -name "*kernel_trace.csv" -delete
head -16 $OUT/trace_summary.txt
timeout 600 python tools/lola_latency.py LoLa --graph > $OUT/lola.txt 2>&1
tail -9 $OUT/lola.txt | cut -c1-300
timeout 900 python tools/cifar_latency.py > $OUT/cifar.txt 2>&1
tail -4 $OUT/cifar.txt | cut -c1-300
