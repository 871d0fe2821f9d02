#!/bin/bash
# round 5, GPU call 2: (1) variants + poison + abi tests on the final descriptors; (2) A/B of the true-extent descriptors against the
# 1 GiB ones (libdemon_hip_unb.so, same sources with -DDEMON_RSRC_UNBOUNDED) on ONE box, interleaved; (3) more hardware queues
out=gpurun_out/r5b; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_variants_gpu.py tests/test_poison_gpu.py tests/test_layers_gpu.py -q -x -p no:cacheprovider > $out/tests.log 2>&1; echo "rc $?" >> $out/tests.log
B="python bench.py --no-cpu-baseline --no-e2e --no-roofline --steps 40 --warmup 5"
q() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), round(d['value_single_lane'],1), d['config']['lanes'], d['config']['lanes_mapping'])"; }
for i in 1 2 3; do
  timeout 300 $B 2>/dev/null | q bounded >> $out/ab.txt
  DEMON_HIP_LIB=$PWD/demon_amd/libdemon_hip_unb.so timeout 300 $B 2>/dev/null | q unbounded >> $out/ab.txt
done
for nq in 8 16; do
  GPU_MAX_HW_QUEUES=$nq timeout 300 $B 2>/dev/null | tail -1 > $out/queues_$nq.json
  python -c "import json; d=json.load(open('$out/queues_$nq.json')); print('GPU_MAX_HW_QUEUES=$nq', round(d['value'],1), round(d['value_single_lane'],1), d['config']['lanes'], d['config']['lanes_calibration_pairs_per_s'])" >> $out/ab.txt
done
GPU_MAX_HW_QUEUES=8 timeout 300 python bench.py --no-cpu-baseline --no-e2e --no-roofline --steps 40 --warmup 5 --lanes 4 2>/dev/null | q "queues8_lanes4" >> $out/ab.txt
GPU_MAX_HW_QUEUES=8 timeout 300 python bench.py --no-cpu-baseline --no-e2e --no-roofline --steps 40 --warmup 5 --lanes 5 2>/dev/null | q "queues8_lanes5" >> $out/ab.txt
cat $out/ab.txt; tail -3 $out/tests.log
