#!/bin/bash
# round 5, visit y: matrix-core scalar GEMM with scalar gather-table loads and K steps requested two ahead (exact wait counts): words, kernel time A/B, batch time
O=gpurun_out/r05y; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_evaluator.py tests/test_cryptonets_mnist.py tests/test_deferred.py tests/test_lola.py -m gpu -x -q -k "gemm or end_to_end or unchanged or lola or deferred or dense" > $O/pytest.txt 2>&1
tail -2 $O/pytest.txt
L=$PWD/cryptonets_amd/lib
for v in gemmold gemmd1 "" gemmd3; do
  echo "== lib ${v:-default(d2)}"
  CNHIP_LIB=$L/libcnhip${v:+_$v}.so python tools/gemm_probe.py 20 2>&1 | tail -3
done | tee $O/gemm_probe.txt
for t in 4 2; do echo "== default lib, BENCH_CONV_TILE=$t"; BENCH_CONV_TILE=$t python tools/gemm_probe.py 20 2>&1 | tail -3; done | tee -a $O/gemm_probe.txt
for v in gemmold "" gemmold ""; do
  CNHIP_LIB=$L/libcnhip${v:+_$v}.so python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-unchanged-caller --no-single-image --no-relinearize-late 2>> $O/bench.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${v:-new}', d['ms_per_step'], d['value'], d['verified_against_integer_model'])"
done | tee $O/bench_ab.txt
