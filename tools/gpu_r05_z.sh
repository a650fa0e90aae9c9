#!/bin/bash
# round 5, visit z: VALU scalar GEMM of the convolution (2-D grid, weights requested per set, exact tap count, batched epilogue loads) + XCD-contiguous matrix-core GEMM:
# words (whole GEMM-related suite), kernel time A/B against the previous kernels, batch time
O=gpurun_out/r05z; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_evaluator.py tests/test_cryptonets_mnist.py tests/test_deferred.py tests/test_lola.py tests/test_layers.py tests/test_basic_operations.py -m gpu -x -q -k "gemm or end_to_end or unchanged or lola or deferred or dense or Layer or layer or Dense or Sparse or Pool" > $O/pytest.txt 2>&1
tail -2 $O/pytest.txt
L=$PWD/cryptonets_amd/lib
for v in gemmold "" gemmold ""; do
  echo "== lib ${v:-default}"
  CNHIP_LIB=$L/libcnhip${v:+_$v}.so python tools/gemm_probe.py 20 2>&1 | tail -3
done | tee $O/gemm_probe.txt
for v in gemmold "" gemmold ""; do
  CNHIP_LIB=$L/libcnhip${v:+_$v}.so python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-unchanged-caller --no-single-image --no-relinearize-late 2>> $O/bench.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${v:-new}', d['ms_per_step'], d['value'], d['verified_against_integer_model'])"
done | tee $O/bench_ab.txt
