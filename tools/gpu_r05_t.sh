#!/bin/bash
# Round 5, visit T: HBM traffic counters of the batched NTT launch (roofline.traffic) and the SURVEY 8(d) grid on the final tree
O=gpurun_out/r05t; mkdir -p $O
bash tools/pmc_traffic.sh > $O/pmc_traffic.txt 2>&1; tail -12 $O/pmc_traffic.txt
cp gpurun_out/pmc_traffic/ntt_hbm_traffic.json $O/ 2>/dev/null; cp gpurun_out/pmc_traffic/fetch_size_counter_collection.csv $O/ 2>/dev/null; cp gpurun_out/pmc_traffic/write_size_counter_collection.csv $O/ 2>/dev/null
timeout 600 python tools/ntt_grid.py > $O/ntt_grid.txt 2>&1; tail -14 $O/ntt_grid.txt | cut -c1-200
