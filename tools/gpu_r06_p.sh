#!/bin/bash
# Round 6, visit P: the fold kernel with a block per (output, component, limb): parity (deferred suite) and the literal caller, alternating with the build before it
O=gpurun_out/r06p; mkdir -p $O
timeout 900 python -m pytest tests/test_deferred.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
for rep in 1 2 3; do
  python tools/replay_reference_calls.py --trained --threads 1 --literal-threads 16,256 --steps 5 > $O/replay.txt 2> $O/replay.err
  python -c "
import json
for ln in open('$O/replay.txt'):
    d = json.loads(ln); print('rep $rep:', d['caller'][:40], d['threads'], d['ms_per_batch'], d.get('frac_of_batched'), d.get('words_identical'))"
done
export TMPDIR=/tmp; R=$PWD
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $R/$O/prof -- python $R/tools/replay_reference_calls.py --trained --threads 1 --literal-threads 16 --steps 6 > /dev/null 2> $R/$O/prof.err)
KT=$(find $O/prof -name "*kernel_trace.csv" | head -1); python tools/trace_gaps.py $KT 0.3 3 | grep -E "fold|sample|window" | cut -c1-120; find $O/prof -name "*kernel_trace.csv" -delete
