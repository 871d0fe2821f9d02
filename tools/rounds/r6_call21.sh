#!/bin/bash
# round 6, GPU call 21: the silent abort in demon_set_weight again, this time with pytest's faulthandler off so that the preload shim prints the NATIVE
# backtrace of the raising thread; HIP's own error log on (AMD_LOG_LEVEL=1: errors only)
out=gpurun_out/r6v; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
gcc -shared -fPIC -o /tmp/abort_bt.so tools/dbg/abort_bt.c
( time AMD_LOG_LEVEL=1 LD_PRELOAD=/tmp/abort_bt.so timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider -p no:faulthandler ) > $out/gputest_bt.log 2>&1; echo "pytest rc $?" >> $out/gputest_bt.log
grep -n -B5 -A40 "SIGABRT\|SIGSEGV" $out/gputest_bt.log | head -120
tail -4 $out/gputest_bt.log
cat /proc/sys/vm/max_map_count
