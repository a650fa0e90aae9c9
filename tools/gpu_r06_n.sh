#!/bin/bash
# Round 6, visit N: is the 4.5 % of bench.py --stagger the device-side ordering, or the split of the squaring layer into cn_multiply + cn_relinearize?  (BENCH_STAGGER_NOWAIT=1: both halves, no wait)
O=gpurun_out/r06n; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_multi_context.py -m gpu -x -q -k "broadcast" > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-unchanged-caller --no-single-image --no-relinearize-late"
for rep in 1 2; do for mode in "st0" "st1" "st1nowait"; do
  case $mode in st0) E=""; A="--stagger 0";; st1) E=""; A="--stagger 1";; st1nowait) E="BENCH_STAGGER_NOWAIT=1"; A="--stagger 1";; esac
  env $E $B $A > $O/bench_$mode.json 2> $O/bench.err
  python -c "
import json; d=json.loads(open('$O/bench_$mode.json').read().strip().splitlines()[-1])
print('$mode rep $rep:', d['value'], d['ms_per_step'], d['verified_against_integer_model'])"
done; done
