#!/bin/bash
# round 4, visit a: both key-switch conventions on the device (kernels, keygen, coefficient-form keys, start-up self-test) + default bench line (no regression)
OUT=gpurun_out/r04a
mkdir -p $OUT
timeout 1200 python -m pytest tests/test_self_test.py tests/test_gpu_client.py "tests/test_gpu_evaluator.py::test_key_switch_xi_convention_variants_agree" \
  "tests/test_gpu_evaluator.py::test_key_switch_variants_agree" tests/test_lola.py -m gpu -x -q > $OUT/pytest.txt 2>&1
tail -6 $OUT/pytest.txt
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
cut -c1-400 $OUT/bench.json
tail -3 $OUT/bench.err
