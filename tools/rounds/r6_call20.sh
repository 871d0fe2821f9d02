#!/bin/bash
# round 6, GPU call 20: the -m gpu suite as the collection script runs it (TMPDIR=/tmp, after a bench run), under the native-backtrace shim: where does the
# silent abort in demon_set_weight of test_shipped_plan_matches_oracle[192x256_n64] come from?
out=gpurun_out/r6u; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
gcc -shared -fPIC -o /tmp/abort_bt.so tools/dbg/abort_bt.c
ls /lib/x86_64-linux-gnu/libc_malloc_debug.so.0 2>&1 | tail -1
free -g | head -2 > $out/mem.txt
( time LD_PRELOAD=/tmp/abort_bt.so timeout 1500 python -m pytest tests -m gpu -q -rs -p no:cacheprovider ) > $out/gputest_bt.log 2>&1; echo "pytest rc $?" >> $out/gputest_bt.log
grep -n -A30 "SIGABRT" $out/gputest_bt.log | head -60
tail -4 $out/gputest_bt.log
free -g | head -2 >> $out/mem.txt; cat $out/mem.txt
