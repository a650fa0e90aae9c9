#!/bin/bash
# Round 5, visit P: compile-time switches of the transform core on k_keyswitch_pair14 (stage fences, barrier placement, SGPR first-pass roots)
O=gpurun_out/r05p; mkdir -p $O
for m in "" _fence _nopre _nosgpr; do
  echo "== libcnhip$m.so" | tee -a $O/ab.txt
  CNHIP_LIB=$PWD/cryptonets_amd/lib/libcnhip$m.so timeout 300 python tools/ks14_probe.py 5488 ks_pair14=1 2>&1 | grep -v "^N =" | tee -a $O/ab.txt
done
