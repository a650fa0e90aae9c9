#!/bin/bash
# round 5, visit ag: overlapping gather lists merged in pairs by the plan (pair_gather_lists): words, kernel time and batch time with CN_GEMM_PAIR=0 / 1
O=gpurun_out/r05ag; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_evaluator.py tests/test_cryptonets_mnist.py tests/test_deferred.py tests/test_lola.py tests/test_layers.py tests/test_basic_operations.py tests/test_raw_operations.py -m gpu -x -q > $O/pytest.txt 2>&1
tail -2 $O/pytest.txt
for v in 0 1 0 1; do echo "== CN_GEMM_PAIR=$v"; CN_GEMM_PAIR=$v python tools/gemm_probe.py 20 2>&1 | tail -3 | head -1; done | tee $O/gemm_probe.txt
for v in 0 1 0 1; do
  CN_GEMM_PAIR=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-unchanged-caller --no-single-image --no-relinearize-late 2>> $O/bench.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pair=$v', d['ms_per_step'], d['value'], d['verified_against_integer_model'])"
done | tee $O/bench_ab.txt
