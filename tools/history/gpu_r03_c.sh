#!/bin/bash
# Round-3 visit C: ChaCha20 sampler + deferred encryptions + sleeping context lock: tests, then the unchanged-caller table (padded taps skipped
# vs literal zero encryptions) over caller-thread counts up to all host cores, then the default bench line
OUT=gpurun_out/r03c
mkdir -p $OUT
nproc
timeout 900 python -m pytest tests/test_gpu_client.py tests/test_deferred.py tests/test_gpu_serialization.py tests/test_examples.py -m gpu -x -q > $OUT/pytest_new.txt 2>&1
grep -n "passed\|failed\|rror" $OUT/pytest_new.txt | head -20
timeout 900 python tools/replay_reference_calls.py --trained --threads 1,4,8,32,64,256 --literal-threads 1,4,32,256 --steps 5 > $OUT/unchanged_caller_replay.txt 2>&1
cut -c1-260 $OUT/unchanged_caller_replay.txt | tail -14
timeout 900 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python -c "import json; d=json.load(open('$OUT/bench.json')); print(d['value'], d['ms_per_step'], d['verified_against_integer_model'], d['unchanged_caller'])" || tail -20 $OUT/bench.err
