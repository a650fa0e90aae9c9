#!/bin/bash
# round 4, visit j: the unchanged CryptoNets caller with the scalar product's own result released at once (bias fold kept) and everything else in batches;
# host time of the queue flushes (CN_DEFER_TRACE=2)
OUT=gpurun_out/r04j
mkdir -p $OUT
CN_DEFER_TRACE=2 python tools/replay_reference_calls.py --trained --threads 16 --literal-threads 16 --steps 2 2> $OUT/flush_times.txt > $OUT/replay.txt
cut -c1-200 $OUT/replay.txt
grep "host time" $OUT/flush_times.txt | tail -14 | cut -c1-200
CN_DEFER_TRACE=1 python tools/replay_reference_calls.py --trained --threads 16 --literal-threads 16 --steps 1 2> $OUT/defer_trace.txt > /dev/null; tail -22 $OUT/defer_trace.txt | cut -c1-160
for i in 1 2; do
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-single-image --no-relinearize-late > $OUT/bench$i.json 2>> $OUT/bench.err
python -c "import json; d=json.loads(open('$OUT/bench$i.json').read().strip().splitlines()[-1]); u=d['unchanged_caller']; print(d['ms_per_step'], d['verified_against_integer_model'], 'unchanged', u['ms_per_step'], u['frac_of_batched'], u['frac_of_batched_mean_over_mean'], u['verified_against_integer_model'], u['windows_ms'])"
done
