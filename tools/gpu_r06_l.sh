#!/bin/bash
# Round 6, visit L: the encryption kernel with a block per (ciphertext, component, limb) (k_encrypt_split, two workgroups per CU) against k_encrypt_fused and the three-launch chain
O=gpurun_out/r06l; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
timeout 600 python -m pytest tests/test_gpu_client.py -m gpu -x -q -k "fused_encryption or encrypt" > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
python tools/encrypt_probe.py 2>&1 | tee $O/probe.txt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $R/$O/prof -- python $R/tools/encrypt_probe.py > /dev/null 2> $R/$O/prof.err)
KT=$(find $O/prof -name "*kernel_trace.csv" | head -1); python tools/summarize_trace.py $KT > $O/trace.txt 2>&1; find $O/prof -name "*kernel_trace.csv" -delete
grep -E "encrypt|sample|expand|ntt_rr" $O/trace.txt | cut -c1-130
