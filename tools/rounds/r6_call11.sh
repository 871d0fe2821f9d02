#!/bin/bash
# round 6, GPU call 11: the -m gpu suite again (the collection's run aborted inside demon_set_weight of test_shipped_plan_matches_oracle[192x256_n64]
# without a message): the plan tests alone first, then everything, stderr kept apart
out=gpurun_out/r6k; mkdir -p $out
cd $GRAFT_REPO_ROOT
( time timeout 1200 python -m pytest tests/test_plans_gpu.py -q -p no:cacheprovider ) > $out/plans.log 2> $out/plans.err; echo "rc $?" >> $out/plans.log
tail -5 $out/plans.log; tail -5 $out/plans.err
( time timeout 2400 python -m pytest tests -m gpu -q -rs -p no:cacheprovider ) > $out/gputest.log 2> $out/gputest.err; echo "pytest rc $?" >> $out/gputest.log
tail -12 $out/gputest.log; tail -8 $out/gputest.err
dmesg 2>/dev/null | tail -5
