#!/bin/bash
# Round 5, visit K: scalar products of one flush merged into one launch per term count - deferred-queue tests, then A/B of the unchanged caller
O=gpurun_out/r05k; mkdir -p $O
timeout 900 python -m pytest tests/test_deferred.py tests/test_call_trace.py tests/test_cryptonets_mnist.py -m gpu -q -x > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for rep in 1 2; do for m in 0 1; do
  echo "== CN_DEFER_MERGE_GEMM=$m" | tee -a $O/ab.txt
  CN_DEFER_MERGE_GEMM=$m python tools/replay_reference_calls.py --trained --threads 16 --literal-threads 16,256 --steps 8 2>/dev/null | cut -c1-260 | tee -a $O/ab.txt
done; done
