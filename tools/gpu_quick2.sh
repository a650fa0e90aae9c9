#!/bin/bash
cd /tmp; rocprofv3 -L 2>/dev/null | grep -iE "^\s*(Name|.*SQ_(WAIT|INST_LEVEL|ACTIVE|BUSY|WAVE|INSTS|LDS|INST_CYCLES|THREAD|IFETCH|BARRIER|LEVEL))" | head -120 > $GRAFT_REPO_ROOT/gpurun_out/counters.txt; wc -l $GRAFT_REPO_ROOT/gpurun_out/counters.txt; head -100 $GRAFT_REPO_ROOT/gpurun_out/counters.txt | cut -c1-160
