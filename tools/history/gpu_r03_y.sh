#!/bin/bash
CHAIN_PROBE_DUMP=1 python tools/chain_concurrency_probe.py LoLa > gpurun_out/chain_probe.txt 2>&1; grep -v "^    " gpurun_out/chain_probe.txt | tail -12
