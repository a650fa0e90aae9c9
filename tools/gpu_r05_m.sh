#!/bin/bash
# Round 5, visit M: pair14 with the whole-limb digit specialisation: parity, probe, counters of the final shape, CIFAR line
O=gpurun_out/r05m; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
timeout 900 python -m pytest tests/test_gpu_evaluator.py tests/test_lola_cifar.py -m gpu -q -x -k "n16384 or key_switch or fused_rotate_and_add or c5_shapes or cifar" > $O/pytest_ks.txt 2>&1; tail -2 $O/pytest_ks.txt
timeout 300 python tools/ks14_probe.py 5488 ks_pair14=1 ks_pair14=0 2>&1 | tee $O/ks14_probe.txt
P="python $R/tools/ks14_probe.py 5488 ks_pair14=1"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -f csv -d $R/$O/p1 -- $P > $R/$O/run1.txt 2>&1)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS -f csv -d $R/$O/p2 -- $P > $R/$O/run2.txt 2>&1)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_ANY -f csv -d $R/$O/p3 -- $P > $R/$O/run3.txt 2>&1)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $R/$O/p4 -- $P > $R/$O/run4.txt 2>&1)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $R/$O/p5 -- $P > $R/$O/run5.txt 2>&1)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 -f csv -d $R/$O/p6 -- $P > $R/$O/run6.txt 2>&1)
python tools/ks14_counters.py $O $O/ks14_counters.json 45.6 | grep -E "per_wave|bytes_per|SQ_WAVE_CYCLES|SQ_ACTIVE_INST_ANY|SQ_WAIT"
find $O -name "*kernel_trace.csv" -delete
find $O -name "*counter_collection.csv" -size +20M -delete
cp $O/ks14_counters.json profiles/r05_ks14_counters.json
python bench.py --workload cifar --steps 3 --warmup 2 > $O/cifar.json 2> $O/cifar.err; tail -1 $O/cifar.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_image'], d['verified_against_integer_model']); print(json.dumps(d.get('key_switch'))[:1800])"
