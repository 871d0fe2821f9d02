#!/bin/bash
# round 5, GPU call 6: plans tuned with five / six passes in flight against the four-pass plan (the loader takes the plan nearest to
# --plan-lanes), and a fresh latency plan (one pass at a time) for the metric's configuration
out=gpurun_out/r5f; mkdir -p $out
cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --no-e2e --no-roofline --steps 40 --warmup 5"
q() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), round(d['value_single_lane'],1), d['config']['lanes'], d['config']['lanes_mapping'])"; }
timeout 300 $B --plan-lanes 4 2>/dev/null | q "plan l4" >> $out/ab.txt
for L in 5 6; do
  ( time timeout 900 python tools/tune.py --batch 32 --lanes $L --rounds 3 --outdir $out ) > $out/tune_n32_l$L.log 2>&1
  cp $out/plan_192x256_n32_l$L.json demon_amd/tuned/
  timeout 300 $B --plan-lanes $L 2>/dev/null | q "plan l$L" >> $out/ab.txt
  timeout 300 $B --plan-lanes $L 2>/dev/null | q "plan l$L" >> $out/ab.txt
done
timeout 300 $B --plan-lanes 4 2>/dev/null | q "plan l4" >> $out/ab.txt
( time timeout 900 python tools/tune.py --batch 32 --lanes 1 --rounds 3 --outdir $out ) > $out/tune_n32_l1.log 2>&1
cp demon_amd/tuned/plan_192x256_n32.json $out/plan_192x256_n32_round4.json
cp $out/plan_192x256_n32.json demon_amd/tuned/
timeout 300 $B --plan-lanes 4 2>/dev/null | q "plan l4 + new latency plan" >> $out/ab.txt
cat $out/ab.txt
