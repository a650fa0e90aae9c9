#!/bin/bash
# Round-3 visit E: idle-triggered flush policy: deferred tests, unchanged-caller table, default bench (fair CPU baseline + host compute probe),
# bench --workload cifar (verification through the big dense layer) and --workload lola
OUT=gpurun_out/r03e
mkdir -p $OUT
cat /sys/fs/cgroup/cpu.max 2>/dev/null; lscpu | grep -i "model name\|socket\|core(s)\|thread(s)\|numa node(s)" | head -6
timeout 600 python -m pytest tests/test_deferred.py tests/test_cryptonets_mnist.py tests/test_layers.py -m gpu -x -q 2>&1 | grep -n "passed\|failed\|rror" | head
timeout 900 python tools/replay_reference_calls.py --trained --threads 1,4,32,256 --literal-threads 1,4,32,256 --steps 5 > $OUT/unchanged_caller_replay.txt 2>&1
cut -c1-250 $OUT/unchanged_caller_replay.txt | tail -10
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
python -c "import json; d=json.load(open('$OUT/bench.json')); print(d['value'], d['ms_per_step'], d['verified_against_integer_model'], d['unchanged_caller']); print(d['cpu_baseline'])" || tail -20 $OUT/bench.err
timeout 900 python bench.py --workload lola --steps 10 --warmup 2 > $OUT/bench_lola.json 2> $OUT/bench_lola.err; cut -c1-300 $OUT/bench_lola.json
timeout 1200 python bench.py --workload cifar --steps 2 --warmup 1 > $OUT/bench_cifar.json 2> $OUT/bench_cifar.err; cut -c1-400 $OUT/bench_cifar.json; tail -3 $OUT/bench_cifar.err
