#!/bin/bash
# Round 6, visit T: what the device runs for the literal unchanged caller, kernel by kernel WITHOUT overlap (AMD_SERIALIZE_KERNEL=3: the runtime waits around every launch), next to the
# batched program's serialised trace (profiles/r06_bench_kernel_trace_summary.txt)
O=gpurun_out/r06t; mkdir -p $O
export TMPDIR=/tmp; R=$PWD
(cd /tmp && AMD_SERIALIZE_KERNEL=3 timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $R/$O/prof -- python $R/tools/replay_reference_calls.py --trained --threads 1 --literal-threads 16 --steps 6 --warmup 1 > $R/$O/run.txt 2> $R/$O/prof.err)
KT=$(find $O/prof -name "*kernel_trace.csv" | head -1); python tools/trace_gaps.py $KT 0.25 2 > $O/literal_serialised.txt 2>&1; find $O/prof -name "*kernel_trace.csv" -delete
cut -c1-130 $O/literal_serialised.txt | head -40; tail -2 $O/run.txt | cut -c1-200
