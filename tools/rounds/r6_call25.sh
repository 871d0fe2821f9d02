#!/bin/bash
# round 6, GPU call 25: the RCCL loader (already-loaded copy first, RTLD_LOCAL): the soak that ended in "double free or corruption" at exit, shortened; the
# distributed / bench tests
out=gpurun_out/r6z; mkdir -p $out
cd $GRAFT_REPO_ROOT
( time timeout 300 python -X faulthandler tools/dbg/soak_set_weights.py --rccl --spawn --seconds 40 ) > $out/soak.log 2>&1; echo "rc $?" >> $out/soak.log
grep -v "^round\|^  File" $out/soak.log | tail -12
( time timeout 900 python -m pytest tests/test_distributed.py tests/test_bench_gpu.py -q -m gpu ) > $out/tests.log 2>&1; echo "rc $?" >> $out/tests.log
tail -5 $out/tests.log
