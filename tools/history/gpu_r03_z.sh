#!/bin/bash
# visit Z: a quarter of the compute units per context (CU-masked streams)?
for m in 0 1 2; do echo "CN_CU_MASK=$m"; CN_CU_MASK=$m python tools/chain_concurrency_probe.py LoLa 2>&1 | grep -v "^contexts \[[123]\] " | tail -7; done
