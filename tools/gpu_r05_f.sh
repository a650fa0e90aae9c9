#!/bin/bash
# Round 5, visit F: counters of k_keyswitch_pair14 (final shape), the CIFAR line with its key_switch block, kernel trace
O=gpurun_out/r05f; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
timeout 900 python -m pytest tests/test_gpu_evaluator.py tests/test_lola_cifar.py -m gpu -q -x -k "n16384 or key_switch or fused_rotate_and_add or c5_shapes or cifar" > $O/pytest_ks.txt 2>&1; tail -3 $O/pytest_ks.txt
rocprofv3 -L 2>/dev/null | grep -i -E "F64|INSTS_VALU" | head -40 > $O/counters_avail.txt; head -30 $O/counters_avail.txt | cut -c1-160
P="python $R/tools/ks14_probe.py 5488 ks_pair14=1"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -f csv -d $R/$O/p1 -- $P > $R/$O/run1.txt 2>&1)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS -f csv -d $R/$O/p2 -- $P > $R/$O/run2.txt 2>&1)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_ANY -f csv -d $R/$O/p3 -- $P > $R/$O/run3.txt 2>&1)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $R/$O/p4 -- $P > $R/$O/run4.txt 2>&1)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $R/$O/p5 -- $P > $R/$O/run5.txt 2>&1)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 -f csv -d $R/$O/p6 -- $P > $R/$O/run6.txt 2>&1)
tail -2 $O/run6.txt | cut -c1-200
python tools/ks14_counters.py $O $O/ks14_counters.json 45.6 | tail -32
find $O -name "*kernel_trace.csv" -delete
find $O -name "*counter_collection.csv" -size +20M -delete
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $R/$O/prof -- python $R/bench.py --workload cifar --steps 3 --warmup 2 > $R/$O/cifar_prof.json 2> $R/$O/cifar_prof.err)
KT=$(find $O/prof -name "*kernel_trace.csv" | head -1); python tools/summarize_trace.py $KT > $O/cifar_trace_summary.txt 2>&1; find $O/prof -name "*kernel_trace.csv" -delete
head -14 $O/cifar_trace_summary.txt | cut -c1-140
python bench.py --workload cifar --steps 3 --warmup 2 > $O/cifar.json 2> $O/cifar.err; tail -1 $O/cifar.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_image'], d['verified_against_integer_model']); print(json.dumps(d.get('key_switch'))[:1500])"
