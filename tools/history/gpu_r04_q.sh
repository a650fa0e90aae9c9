#!/bin/bash
# round 4, visit q: the literal unchanged caller at 64 / 256 caller threads - merged calls on / off x the bounded spin of a timed-out sleeper on / off (3 measurements each, warm-up 2)
OUT=gpurun_out/r04q
mkdir -p $OUT
for s in 1 0; do for m in "" "--no-merged"; do
  echo -n "timeout_spin=$s merged=${m:-on}: "
  CN_LOCK_TIMEOUT_SPIN=$s python tools/replay_reference_calls.py --trained --threads 1 --literal-threads 64,64,64,256,256,256 --steps 5 $m 2>/dev/null | python -c "
import json,sys
rows=[json.loads(l) for l in sys.stdin if 'padded' in l]
print(' '.join('%d:%.1f' % (r['threads'], r['ms_per_batch']) for r in rows))"
done; done | tee $OUT/ab.txt
