#!/bin/bash
# Round 6, visit V: the scalar-product kernels of the queued calls, alone on the device (AMD_SERIALIZE_KERNEL=3), with address tables (CN_DEFER_REL=0) / index tables (1) / index tables + pairs
O=gpurun_out/r06v; mkdir -p $O
export TMPDIR=/tmp; R=$PWD
for mode in "0 0" "1 0" "1 1"; do
  set -- $mode
  (cd /tmp && CN_DEFER_REL=$1 CN_DEFER_PAIR=$2 AMD_SERIALIZE_KERNEL=3 timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $R/$O/prof_$1$2 -- python $R/tools/replay_reference_calls.py --trained --threads 1 --literal-threads 16 --steps 6 --warmup 1 > $R/$O/run_$1$2.txt 2> $R/$O/prof_$1$2.err)
  KT=$(find $O/prof_$1$2 -name "*kernel_trace.csv" | head -1); python tools/summarize_trace.py $KT > $O/summary_$1$2.txt 2>&1; find $O/prof_$1$2 -name "*kernel_trace.csv" -delete
  echo "== CN_DEFER_REL=$1 CN_DEFER_PAIR=$2"; grep -E "scalar_gemm|encrypt_fold|sample_small" $O/summary_$1$2.txt | cut -c1-130
done
