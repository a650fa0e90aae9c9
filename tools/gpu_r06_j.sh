#!/bin/bash
# Round 6, visit J: what the half-batch stagger of the two plaintext-prime channels is worth on this tree (bench.py --stagger 0 / 1, alternating)
O=gpurun_out/r06j; mkdir -p $O
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-unchanged-caller --no-single-image --no-relinearize-late"
for rep in 1 2 3; do for st in 0 1; do
  $B --stagger $st > $O/bench_st${st}_$rep.json 2> $O/bench.err
  python -c "
import json; d=json.loads(open('$O/bench_st${st}_$rep.json').read().strip().splitlines()[-1])
print('stagger $st rep $rep:', d['value'], d['ms_per_step'], d['verified_against_integer_model'])"
done; done
