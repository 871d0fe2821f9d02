#!/bin/bash
# round 6, GPU call 15: in-pass tune of the batch-16 latency plan; every shipped plan against the oracle; the throughput-mode plan of the metric's
# configuration refined on the metric itself (tools/refine_plan.py) with the in-pass latency plan and the flat-form re-tune as alternatives; A/B
out=gpurun_out/r6p; mkdir -p $out
cd $GRAFT_REPO_ROOT
( time timeout 600 python tools/inpass_tune.py --batch 16 --rounds 2 --out $out/plan_192x256_n16.json --verify 60 ) > $out/inpass_n16.log 2>&1
grep -E "^round|verify" $out/inpass_n16.log
[ -s $out/plan_192x256_n16.json ] && cp $out/plan_192x256_n16.json demon_amd/tuned/plan_192x256_n16.json
( time timeout 1500 python -m pytest tests/test_plans_gpu.py -q -p no:cacheprovider ) > $out/plans.log 2>&1; echo "rc $?" >> $out/plans.log
tail -5 $out/plans.log
( time timeout 2400 python tools/refine_plan.py --base demon_amd/tuned/plan_192x256_n32_l4.json --alt demon_amd/tuned/plan_192x256_n32.json --alt gpurun_in/alt_l4_r6m.json --alt gpurun_in/alt_l4_r6g.json --out $out/plan_192x256_n32_l4.json ) > $out/refine.log 2>&1
tail -30 $out/refine.log
q() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), round(d['value_single_lane'],1), d['config']['lanes'], round(d.get('value_image_features_hoisted') or 0,1))"; }
B="python bench.py --no-cpu-baseline --no-e2e --no-roofline --steps 60 --warmup 5"
cp demon_amd/tuned/plan_192x256_n32_l4.json $out/plan_192x256_n32_l4_before.json
for rep in 1 2 3; do
  cp $out/plan_192x256_n32_l4_before.json demon_amd/tuned/plan_192x256_n32_l4.json
  timeout 300 $B 2>/dev/null | q "shipped l4 plan" >> $out/ab.txt
  [ -s $out/plan_192x256_n32_l4.json ] && cp $out/plan_192x256_n32_l4.json demon_amd/tuned/plan_192x256_n32_l4.json
  timeout 300 $B 2>/dev/null | q "refined l4 plan" >> $out/ab.txt
done
cat $out/ab.txt
