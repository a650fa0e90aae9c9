#!/bin/bash
# Round-3 visit F: heavy-depth layer-boundary flush; call-trace recorder / player; LoLa unchanged caller; unchanged-caller table; default bench
OUT=gpurun_out/r03f
mkdir -p $OUT
timeout 900 python -m pytest tests/test_call_trace.py tests/test_deferred.py tests/test_lola.py -m gpu -x -q 2>&1 | grep -n "passed\|failed\|rror" | head
timeout 900 python tools/lola_unchanged_caller.py LoLa --reps 20 > $OUT/lola_unchanged_caller.txt 2>&1; cut -c1-260 $OUT/lola_unchanged_caller.txt | tail -10
timeout 900 python tools/replay_reference_calls.py --trained --threads 1,4,16,64,256 --literal-threads 1,4,16,64,256 --steps 5 > $OUT/unchanged_caller_replay.txt 2>&1
cut -c1-250 $OUT/unchanged_caller_replay.txt | tail -12
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
python -c "import json; d=json.load(open('$OUT/bench.json')); print(d['value'], d['ms_per_step'], d['verified_against_integer_model'], d['unchanged_caller']); c=d['cpu_baseline']; print({k:c[k] for k in c if k!='sample'})" || tail -20 $OUT/bench.err
