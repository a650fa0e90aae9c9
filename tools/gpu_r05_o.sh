#!/bin/bash
# Round 5, visit O: the compiler's scheduling strategy for the key-switch translation unit, and the forward recentring schedule, on k_keyswitch_pair14
O=gpurun_out/r05o; mkdir -p $O
for m in "" _sdefault _smaxilp _sminreg _smaxocc _recall; do
  echo "== libcnhip$m.so" | tee -a $O/ab.txt
  CNHIP_LIB=$PWD/cryptonets_amd/lib/libcnhip$m.so timeout 300 python tools/ks14_probe.py 5488 ks_pair14=1 ks_pair14=0 2>&1 | grep -v "^N =" | tee -a $O/ab.txt
done
