#!/bin/bash
# visit U: LoLa latency eager / recorded with a hardware queue per context; kernel trace of one eager run
O=gpurun_out/r03u; mkdir -p $O
python tools/lola_latency.py LoLa --graph 2>/dev/null | tail -12
CN_STREAM_PROBE=0 python tools/lola_latency.py LoLa --graph 2>/dev/null | tail -4
