#!/bin/bash
# HBM traffic of the batched NTT launch (roofline.traffic): two separate counter passes, then the calibrated summary.
OUT=gpurun_out/pmc_traffic
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
(cd /tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $R/$OUT/fetch -- python $R/tools/pmc_traffic.py > $R/$OUT/fetch.txt 2>&1)
(cd /tmp && rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $R/$OUT/write -- python $R/tools/pmc_traffic.py > $R/$OUT/write.txt 2>&1)
F=$(find $OUT/fetch -name "*counter_collection.csv" | head -1); W=$(find $OUT/write -name "*counter_collection.csv" | head -1)
cp $F $OUT/fetch_size_counter_collection.csv; cp $W $OUT/write_size_counter_collection.csv
python tools/pmc_summarize.py $F $W $OUT/ntt_hbm_traffic.json | tail -30
find $OUT -name "*kernel_trace.csv" -delete
