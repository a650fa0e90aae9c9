#!/bin/bash
# Round 6, visit AZ: the second stream of a context chosen at creation, off every live context's queues where a queue is left - tests, the parts again, LoLa (four contexts)
R=$(pwd); O=$R/gpurun_out/r06az; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_multi_context.py tests/test_lola.py tests/test_gpu_evaluator.py -q -m gpu -x 2>&1 | tail -3 | tee $O/test.txt
for rep in 1 2; do
  for cfg in "0 2" "0 3" "0 4" "1 3"; do set -- $cfg
    CN_SQ_PARTS=$2 python bench.py --stagger $1 --steps 40 --warmup 3 --no-cpu-baseline --no-single-image --no-relinearize-late --no-unchanged-caller 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('stagger $1 parts $2 rep $rep:', d['value'], d['ms_per_step'], d['verified_against_integer_model'])" | tee -a $O/ab.txt
  done
done
python bench.py --workload lola --steps 20 --warmup 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lola', d['value'], d.get('verified'), d.get('config', {}).get('stream_tries'))" | tee -a $O/ab.txt
