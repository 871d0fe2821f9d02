#!/bin/bash
# round 6, GPU call 24: the silent abort, narrowed: a soak of create / set_weights / pass / close with and without what tests/test_distributed.py does first
out=gpurun_out/r6y; mkdir -p $out
cd $GRAFT_REPO_ROOT
( time timeout 600 python -X faulthandler tools/dbg/soak_set_weights.py --rccl --spawn --seconds 400 ) > $out/soak_rccl_spawn.log 2>&1; echo "rc $?" >> $out/soak_rccl_spawn.log
tail -4 $out/soak_rccl_spawn.log


