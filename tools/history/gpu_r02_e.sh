#!/bin/bash
# Round-2 visit E: new client / recording tests, key-switch micro-benchmark (pipelined + priority), smoke, forced-distributed CryptoNets
# bench, kernel traces of LoLa-MNIST (eager and recorded) and of the LoLa-CIFAR shapes, HBM traffic of the batched NTT (roofline.traffic)
OUT=gpurun_out/r02e
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.txt 2>&1
tail -5 $OUT/pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | cut -c1-200
timeout 300 ./tools/ubench_ks > $OUT/ubench_ks.txt 2>&1
head -6 $OUT/ubench_ks.txt
BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-unchanged-caller > $OUT/bench_forcedist.json 2> $OUT/bench_forcedist.err
echo "forced dist:"; cut -c1-200 $OUT/bench_forcedist.json; tail -1 $OUT/bench_forcedist.err | cut -c1-160
export TMPDIR=/tmp
R=$PWD
(cd /tmp && rocprofv3 --kernel-trace --stats -f csv -d $R/$OUT/lola_eager -- python $R/tools/lola_latency.py LoLa > $R/$OUT/lola_eager.txt 2>&1)
KT=$(find $OUT/lola_eager -name "*kernel_trace.csv" | head -1); python tools/summarize_trace.py $KT > $OUT/lola_eager_trace_summary.txt 2>&1
(cd /tmp && rocprofv3 --kernel-trace --stats -f csv -d $R/$OUT/lola_graph -- python $R/tools/lola_latency.py LoLa --graph > $R/$OUT/lola_graph.txt 2>&1)
KT=$(find $OUT/lola_graph -name "*kernel_trace.csv" | head -1); python tools/summarize_trace.py $KT > $OUT/lola_graph_trace_summary.txt 2>&1
tail -7 $OUT/lola_graph.txt | cut -c1-220
head -12 $OUT/lola_eager_trace_summary.txt | cut -c1-130
(cd /tmp && rocprofv3 --kernel-trace --stats -f csv -d $R/$OUT/cifar -- python $R/tools/cifar_latency.py > $R/$OUT/cifar.txt 2>&1)
KT=$(find $OUT/cifar -name "*kernel_trace.csv" | head -1); python tools/summarize_trace.py $KT > $OUT/cifar_trace_summary.txt 2>&1
tail -3 $OUT/cifar.txt | cut -c1-220; head -12 $OUT/cifar_trace_summary.txt | cut -c1-130
find $OUT -name "*kernel_trace.csv" -delete
# HBM traffic of the batched NTT launch: separate FETCH_SIZE / WRITE_SIZE passes, calibrated on cn_add (tools/pmc_traffic.py)
(cd /tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $R/$OUT/fetch -- python $R/tools/pmc_traffic.py > $R/$OUT/fetch.txt 2>&1)
(cd /tmp && rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $R/$OUT/write -- python $R/tools/pmc_traffic.py > $R/$OUT/write.txt 2>&1)
F=$(find $OUT/fetch -name "*counter_collection.csv" | head -1); W=$(find $OUT/write -name "*counter_collection.csv" | head -1)
cp $F $OUT/fetch_size_counter_collection.csv; cp $W $OUT/write_size_counter_collection.csv
python tools/pmc_summarize.py $F $W $OUT/ntt_hbm_traffic.json | grep -i "traffic_over\|correction" | head
find $OUT -name "*kernel_trace.csv" -delete
