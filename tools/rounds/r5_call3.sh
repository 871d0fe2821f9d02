#!/bin/bash
# round 5, GPU call 3: runtime knobs under lanes -- hardware queues (GPU_MAX_HW_QUEUES), kernel arguments in device memory
# (HIP_FORCE_DEV_KERNARG), lane counts up to 6 -- and the extents A/B again after hoisting the extent arithmetic out of the K loops
out=gpurun_out/r5c; mkdir -p $out
cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --no-e2e --no-roofline --steps 40 --warmup 5"
q() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), round(d['value_single_lane'],1), d['config']['lanes'], d['config']['lanes_mapping'], d['config']['lanes_calibration_pairs_per_s'])"; }
for i in 1 2; do
  timeout 300 $B 2>/dev/null | q bounded >> $out/ab.txt
  DEMON_HIP_LIB=$PWD/demon_amd/libdemon_hip_unb.so timeout 300 $B 2>/dev/null | q unbounded >> $out/ab.txt
done
for nq in 5 6 8 10 12; do
  GPU_MAX_HW_QUEUES=$nq timeout 300 $B --max-lanes 6 2>/dev/null | q "queues$nq" >> $out/ab.txt
done
HIP_FORCE_DEV_KERNARG=1 timeout 300 $B 2>/dev/null | q "devkernarg" >> $out/ab.txt
HIP_FORCE_DEV_KERNARG=1 GPU_MAX_HW_QUEUES=8 timeout 300 $B --max-lanes 6 2>/dev/null | q "devkernarg_queues8" >> $out/ab.txt
GPU_MAX_HW_QUEUES=8 timeout 300 $B --max-lanes 6 2>/dev/null | q "queues8_again" >> $out/ab.txt
cat $out/ab.txt
