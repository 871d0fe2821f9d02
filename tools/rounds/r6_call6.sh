#!/bin/bash
# round 6, GPU call 6: the k x 1 (AXIS 0) packed epilogue with its values pinned before the stores -- the variant tests first, then the whole -m gpu
# suite on this build, then the headline A/B against the build without packed epilogues (gpurun_in/libdemon_hip_prev.so)
out=gpurun_out/r6f; mkdir -p $out
cd $GRAFT_REPO_ROOT
( time timeout 900 python -m pytest tests/test_variants_gpu.py -q -p no:cacheprovider ) > $out/tests_variants.log 2>&1
tail -6 $out/tests_variants.log
q() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), round(d['value_single_lane'],1), d['config']['lanes'], d['single_lane']['lane0_outputs_rel_l1'])"; }
B="python bench.py --no-cpu-baseline --no-e2e --no-roofline --steps 60 --warmup 5"
for rep in 1 2; do
  DEMON_HIP_LIB=$PWD/gpurun_in/libdemon_hip_prev.so timeout 300 $B 2>/dev/null | q "previous build" >> $out/ab.txt
  timeout 300 $B 2>/dev/null | q "packed epilogues" >> $out/ab.txt
done
cat $out/ab.txt
( time timeout 2400 python -m pytest tests -m gpu -q -rs -p no:cacheprovider ) > $out/gputest.log 2>&1
tail -15 $out/gputest.log
