#!/bin/bash
# Round 6, visit R: k_keyswitch_pair14 with non-temporal hints on the closing step (libcnhip_nt<mask>.so, tools/build_ks14_nt.py): parity of one variant, then ms per 5488-ciphertext link, alternating
O=gpurun_out/r06r; mkdir -p $O
CNHIP_LIB=$PWD/cryptonets_amd/lib/libcnhip_nt15.so timeout 900 python -m pytest tests/test_gpu_evaluator.py -m gpu -x -q -k "n16384 or fused_rotate_and_add" > $O/pytest_nt15.txt 2>&1; tail -2 $O/pytest_nt15.txt
for rep in 1 2; do for tag in "" _nt1 _nt3 _nt4 _nt15; do
  lib=$PWD/cryptonets_amd/lib/libcnhip$tag.so
  echo "== build '$tag' rep $rep"; CNHIP_LIB=$lib timeout 300 python tools/ks14_probe.py 5488 ks_pair14=1 2>&1 | grep -E "ms" | tail -2 | cut -c1-200
done; done
