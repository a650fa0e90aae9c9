#!/bin/bash
# Round 5, visit C: k_keyswitch_pair14 with the next digit's source words requested under the current digit's passes + the closing step's loads in front of its stores
O=gpurun_out/r05c; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_evaluator.py -m gpu -q -x -k "n16384 or key_switch or fused_rotate_and_add or c5_shapes" > $O/pytest_ks.txt 2>&1; tail -3 $O/pytest_ks.txt
for m in "" _nopf _dbg7 _dbg15 _dbg16 _dbg48 _dbg63; do
  echo "== libcnhip$m.so" | tee -a $O/ks14_dbg.txt
  CNHIP_LIB=$PWD/cryptonets_amd/lib/libcnhip$m.so timeout 300 python tools/ks14_probe.py 5488 ks_pair14=1,ks_chain=1 ks_pair14=1,ks_chain=1,ks_xcd=1 2>&1 | grep -v "^N =" | tee -a $O/ks14_dbg.txt
done
python bench.py --workload cifar --steps 2 --warmup 1 > $O/cifar.json 2> $O/cifar.err; tail -1 $O/cifar.json | cut -c1-300
CN_KS_XCD=1 python bench.py --workload cifar --steps 2 --warmup 1 > $O/cifar_xcd1.json 2> $O/cifar_xcd1.err; tail -1 $O/cifar_xcd1.json | cut -c1-300
