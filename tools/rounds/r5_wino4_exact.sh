#!/bin/bash
# round 5: the mask-free epilogue of conv_wino4 (scalar store offsets on exact-fit grids): parity of every variant (plain and poisoned),
# per-layer A/B (DEMON_WINO4_EXACT=0 = the masked epilogue), end-to-end A/B
out=gpurun_out/r5i; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_variants_gpu.py tests/test_poison_gpu.py tests/test_plans_gpu.py -q -x -p no:cacheprovider -k "four_outputs or whole_net or shipped_plan" > $out/tests.log 2>&1; echo "rc $?" >> $out/tests.log
for spec in "conv3y:32 64 48 64 128 5 1 2 1:16,4,1" "conv3y_v8:32 64 48 64 128 5 1 2 1:16,8,1" "conv3x:32 128 24 64 128 1 5 1 2:16,4,1" "conv2_1y:32 64 48 64 64 3 1 1 1:16,8,1" "conv2_1x:32 64 48 64 64 1 3 1 1:16,8,1" "conv3_1y:32 128 24 32 128 3 1 1 1:16,8,1" "conv4_1y:32 256 12 16 256 3 1 1 1:16,6,1" "conv4x:32 256 12 32 256 1 5 1 2:16,4,1"; do
  name=${spec%%:*}; rest=${spec#*:}; shape=${rest%%:*}; plan=${rest#*:}
  a=$(DEMON_WINO4_EXACT=0 python tools/plan_probe.py conv $shape $plan | tail -1)
  b=$(python tools/plan_probe.py conv $shape $plan | tail -1)
  echo "$name masked: $a | exact: $b" >> $out/layers.txt
done
B="python bench.py --no-cpu-baseline --no-e2e --no-roofline --steps 40 --warmup 5"
q() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), round(d['value_single_lane'],1), d['config']['lanes'])"; }
for i in 1 2; do
  DEMON_WINO4_EXACT=0 timeout 300 $B 2>/dev/null | q masked >> $out/ab.txt
  timeout 300 $B 2>/dev/null | q exact >> $out/ab.txt
done
cat $out/layers.txt $out/ab.txt; tail -3 $out/tests.log
