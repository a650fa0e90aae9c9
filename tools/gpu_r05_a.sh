#!/bin/bash
# Round 5, visit A: the one-launch N = 16384 key switch - word parity of every variant, then the A/B probe, a kernel trace of the CIFAR line (old and new path)
O=gpurun_out/r05a; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
timeout 900 python -m pytest tests/test_gpu_evaluator.py -m gpu -q -x -k "n16384 or key_switch or fused_rotate_and_add or c5_shapes or sum_slots" > $O/pytest_ks.txt 2>&1; tail -5 $O/pytest_ks.txt
timeout 600 python tools/ks14_probe.py 5488 > $O/ks14_probe.txt 2>&1; cat $O/ks14_probe.txt
for v in 0 1; do
  (cd /tmp && CN_KS_PAIR14=$v timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $R/$O/prof$v -- python $R/bench.py --workload cifar --steps 2 --warmup 1 > $R/$O/cifar$v.json 2> $R/$O/cifar$v.err)
  KT=$(find $O/prof$v -name "*kernel_trace.csv" | head -1); python tools/summarize_trace.py $KT > $O/cifar_trace_summary_pair$v.txt 2>&1
  find $O/prof$v -name "*kernel_trace.csv" -delete
  head -12 $O/cifar_trace_summary_pair$v.txt | cut -c1-140
  tail -1 $O/cifar$v.json | cut -c1-400
done
