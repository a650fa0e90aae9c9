#!/bin/bash
O=gpurun_out/r05q; mkdir -p $O
timeout 600 python tools/cifar_noise_trail.py 8 2>&1 | tee $O/noise8.txt | tail -14
timeout 600 python tools/cifar_noise_trail.py 9 2>&1 | tee $O/noise9.txt | tail -12
