#!/bin/bash
# round 5, visit ab: k_keyswitch_pair14 with the prefetch hooks on every path (the last digit requests itself again) against the conditional hooks: words, link time
O=gpurun_out/r05ab; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_evaluator.py tests/test_lola_cifar.py -m gpu -q -x -k "n16384 or c5 or cifar" > $O/pytest_ks.txt 2>&1; tail -2 $O/pytest_ks.txt
for m in _ks14cond "" _ks14cond ""; do
  echo "== libcnhip$m.so"
  CNHIP_LIB=$PWD/cryptonets_amd/lib/libcnhip$m.so timeout 300 python tools/ks14_probe.py 5488 ks_pair14=1,ks_chain=1,ks_xcd=1 2>&1 | grep -v "^N =" | tail -2
done | tee $O/ks14_ab.txt
