#!/bin/bash
# Round-2 visit N: global (not flat) memory instructions for table-derived addresses: full suite, bench, replay table, trace
OUT=gpurun_out/r02n
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.txt 2>&1
tail -3 $OUT/pytest.txt
timeout 600 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python -c "import json; d=json.load(open('$OUT/bench.json')); print(d['value'], d['ms_per_step'], d['key_switch']['ms_per_launch'], d['unchanged_caller'])"
timeout 600 python tools/replay_reference_calls.py --threads 1,4,8,16,32,64 --immediate --trained > $OUT/unchanged_caller_replay.txt 2>&1; cut -c1-150 $OUT/unchanged_caller_replay.txt | tail -9
timeout 300 python tools/lola_latency.py LoLa --graph 2>&1 | tail -3 | cut -c1-150
export TMPDIR=/tmp
R=$PWD
(cd /tmp && rocprofv3 --kernel-trace --stats -f csv -d $R/$OUT/prof -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --serialize > /dev/null 2> $R/$OUT/prof.err)
KT=$(find $OUT/prof -name "*kernel_trace.csv" | head -1); python tools/summarize_trace.py $KT > $OUT/trace.txt 2>&1; find $OUT/prof -name "*kernel_trace.csv" -delete
head -24 $OUT/trace.txt | cut -c1-125
