#!/bin/bash
# Round 6, visit Y: the noise sampler by inversion of the cumulative distribution (one 64-bit word per coefficient, 19 thresholds) instead of Box-Muller in FP64: client-side parity / statistics tests,
# the deferred suite, then the encryption probe under a kernel trace (sampler kernel times)
O=gpurun_out/r06y; mkdir -p $O
export TMPDIR=/tmp; R=$PWD
timeout 1200 python -m pytest tests/test_gpu_client.py tests/test_deferred.py tests/test_cryptotracker.py tests/test_basic_operations.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
python tools/encrypt_probe.py 2>&1 | tee $O/probe.txt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $R/$O/prof -- python $R/tools/encrypt_probe.py > /dev/null 2> $R/$O/prof.err)
KT=$(find $O/prof -name "*kernel_trace.csv" | head -1); python tools/summarize_trace.py $KT > $O/trace.txt 2>&1; find $O/prof -name "*kernel_trace.csv" -delete
grep -E "encrypt|sample" $O/trace.txt | cut -c1-130
python - <<'PY'
# the distribution the device draws: 64 polynomials of noise through cn_noise_poly of fresh zero encryptions is indirect; take the key noise instead (test_device_keys_have_the_right_structure does) -
# here simply the histogram of e = c0 + c1 s of 32 fresh zero encryptions (N = 8192): mean, std, extremes
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from cryptonets_amd._native import Context
g = Context(8192, 549764251649)
g.keygen(5, galois=False)
ch = g.ct_alloc(32)
g.encrypt(0, 0, ch, 0, 32, seed=77)
e = np.asarray(g.noise_poly(ch, 0, 32)) if hasattr(g, "noise_poly") else None
print("noise_poly available:", e is not None)
PY
