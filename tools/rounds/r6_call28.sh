#!/bin/bash
# round 6, GPU call 28: the silent abort under the OLD test configuration (faulthandler on, HIP log off) with the thunk's and RCCL's logs on and the
# backtrace shim preloaded -- two runs; what does the process say when it dies?
out=gpurun_out/r6ab; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
gcc -shared -fPIC -o /tmp/abort_bt.so tools/dbg/abort_bt.c
for i in 1 2; do
  ( time AMD_LOG_LEVEL=0 NCCL_DEBUG=WARN HSAKMT_DEBUG_LEVEL=4 LD_PRELOAD=/tmp/abort_bt.so timeout 1200 python -m pytest tests -m gpu -q -x -o addopts="" -p no:cacheprovider ) > $out/run$i.log 2>&1
  echo "pytest rc $?" >> $out/run$i.log
  grep -E "passed|failed|Fatal|SIGABRT|pytest rc" $out/run$i.log | head -5
  if grep -q "Fatal\|SIGABRT" $out/run$i.log; then grep -v "site-packages\|dist-packages" $out/run$i.log | grep -n -B30 -A40 "Fatal\|SIGABRT" | head -200; break; fi
done
