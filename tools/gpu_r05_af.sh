#!/bin/bash
# round 5, visit af: one-limb form of the small-weight GEMM kernel (convolution: row sums of |w| times q_max below 2^53): words, kernel time and batch time with the
# form switched off (CN_GEMM_ONE_LIMB=0) and on
O=gpurun_out/r05af; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_evaluator.py tests/test_cryptonets_mnist.py tests/test_deferred.py tests/test_lola.py tests/test_layers.py tests/test_basic_operations.py -m gpu -x -q -k "gemm or end_to_end or unchanged or lola or deferred or dense or Layer or layer or Dense or Sparse or Pool" > $O/pytest.txt 2>&1
tail -2 $O/pytest.txt
for v in 0 1 0 1; do echo "== CN_GEMM_ONE_LIMB=$v"; CN_GEMM_ONE_LIMB=$v python tools/gemm_probe.py 20 2>&1 | tail -3; done | tee $O/gemm_probe.txt
for v in 0 1 0 1; do
  CN_GEMM_ONE_LIMB=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-unchanged-caller --no-single-image --no-relinearize-late 2>> $O/bench.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('one_limb=$v', d['ms_per_step'], d['value'], d['verified_against_integer_model'])"
done | tee $O/bench_ab.txt
