#!/bin/bash
# round 6, GPU call 22: hunting the silent abort: the -m gpu suite up to three times under the native-backtrace shim (pytest's faulthandler off),
# HIP / thunk error logs on; stops at the first run that aborts
out=gpurun_out/r6w; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
gcc -shared -fPIC -o /tmp/abort_bt.so tools/dbg/abort_bt.c
for i in 1 2 3; do
  ( time AMD_LOG_LEVEL=1 HSAKMT_DEBUG_LEVEL=3 LD_PRELOAD=/tmp/abort_bt.so timeout 1200 python -m pytest tests -m gpu -q -x -p no:cacheprovider -p no:faulthandler ) > $out/run$i.log 2>&1
  rc=$?; echo "pytest rc $rc" >> $out/run$i.log
  tail -3 $out/run$i.log | head -2
  if grep -q "SIGABRT\|SIGSEGV\|Aborted" $out/run$i.log; then grep -n -B8 -A45 "SIGABRT\|SIGSEGV" $out/run$i.log | head -150; break; fi
done
dmesg 2>/dev/null | tail -5
