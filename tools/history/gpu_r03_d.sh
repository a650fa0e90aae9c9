#!/bin/bash
# Round-3 visit D: futex-sleeping context lock + efficient caller pool: the unchanged-caller table again (skipped / literal taps, 1..256 threads),
# then the default bench line with the fair CPU baseline (arena-tuned allocator, >= 1 s per core and layer type)
OUT=gpurun_out/r03d
mkdir -p $OUT
timeout 600 python -m pytest tests/test_deferred.py -m gpu -x -q 2>&1 | grep -n "passed\|failed\|rror" | head
timeout 900 python tools/replay_reference_calls.py --trained --threads 1,4,8,32,64,256 --literal-threads 1,4,32,256 --steps 5 > $OUT/unchanged_caller_replay.txt 2>&1
cut -c1-250 $OUT/unchanged_caller_replay.txt | tail -12
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
python -c "import json; d=json.load(open('$OUT/bench.json')); print(d['value'], d['ms_per_step'], d['verified_against_integer_model'], d['unchanged_caller']); print(d['cpu_baseline'])" || tail -20 $OUT/bench.err
