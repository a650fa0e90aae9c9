#!/bin/bash
# round 5, visit ah: the queued per-ciphertext scalar products (unchanged PoolLayer) paired like a planned GEMM: words, unchanged-caller windows with CN_GEMM_PAIR=0 / 1
O=gpurun_out/r05ah; mkdir -p $O
timeout 900 python -m pytest tests/test_deferred.py tests/test_cryptonets_mnist.py tests/test_layers.py tests/test_lola.py -m gpu -x -q > $O/pytest.txt 2>&1
tail -2 $O/pytest.txt
for v in 0 1 0 1; do
  CN_GEMM_PAIR=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-single-image --no-relinearize-late 2>> $O/bench.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); u=d['unchanged_caller']; print('pair=$v', d['ms_per_step'], d['verified_against_integer_model'], u['frac_of_batched'], u['windows_ms'], u.get('verified'))"
done | tee $O/bench_ab.txt
