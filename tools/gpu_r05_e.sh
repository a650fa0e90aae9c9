#!/bin/bash
# Round 5, visit E: k_keyswitch_pair14 with the previous digit's multiply-accumulate behind the current digit's first pass (its key words requested a pass earlier)
O=gpurun_out/r05e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_evaluator.py -m gpu -q -x -k "n16384 or key_switch or fused_rotate_and_add or c5_shapes" > $O/pytest_ks.txt 2>&1; tail -3 $O/pytest_ks.txt
for m in "" _macnow _dbg7 _dbg32 _dbg48 _dbg63; do
  echo "== libcnhip$m.so" | tee -a $O/ks14_dbg.txt
  CNHIP_LIB=$PWD/cryptonets_amd/lib/libcnhip$m.so timeout 300 python tools/ks14_probe.py 5488 ks_pair14=1,ks_chain=1 ks_pair14=1,ks_chain=1,ks_xcd=1 2>&1 | grep -v "^N =" | tee -a $O/ks14_dbg.txt
done
