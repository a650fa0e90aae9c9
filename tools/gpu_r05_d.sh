#!/bin/bash
# Round 5, visit D: which of the closing step's memory operations cost k_keyswitch_pair14 its 4.6 ms?
O=gpurun_out/r05d; mkdir -p $O
for m in "" _dbg1 _dbg2 _dbg4 _dbg3 _dbg7 _warm; do
  echo "== libcnhip$m.so" | tee -a $O/ks14_dbg.txt
  CNHIP_LIB=$PWD/cryptonets_amd/lib/libcnhip$m.so timeout 300 python tools/ks14_probe.py 5488 ks_pair14=1,ks_chain=1 ks_pair14=1,ks_chain=1,ks_xcd=1 2>&1 | grep -v "^N =" | tee -a $O/ks14_dbg.txt
done
