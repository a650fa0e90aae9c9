#!/bin/bash
# round 5: counters of the scalar-GEMM kernels (convolution on the VALU kernel, dense 845 -> 100 on the matrix cores) over tools/gemm_probe.py
O=gpurun_out/r05gemm; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
P="python $R/tools/gemm_probe.py 4"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_MFMA -f csv -d $R/$O/p1 -- $P > $R/$O/run1.txt 2>&1)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA -f csv -d $R/$O/p2 -- $P > $R/$O/run2.txt 2>&1)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_FMA_F64 -f csv -d $R/$O/p3 -- $P > $R/$O/run3.txt 2>&1)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $R/$O/p4 -- $P > $R/$O/run4.txt 2>&1)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $R/$O/p5 -- $P > $R/$O/run5.txt 2>&1)
for i in 1 2 3 4 5; do tail -2 $O/run$i.txt | cut -c1-160; done
python tools/gemm_counters.py $O $O/gemm_counters.json | tail -120
find $O -name "*kernel_trace.csv" -delete
find $O -name "*counter_collection.csv" -size +8M -delete
find $O -name "*.db" -delete
