#!/bin/bash
CN_KS_WIDE_MAX=400 python tools/sumslots_merge_probe.py 2>&1 | tail -2
python tools/sumslots_merge_probe.py 2>&1 | tail -1
