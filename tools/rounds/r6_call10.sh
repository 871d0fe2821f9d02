#!/bin/bash
# round 6, GPU call 10: the throughput-mode plan refined on the metric itself (tools/refine_plan.py): layer groups switched to the entries of a fresh
# four-replay tune / the latency plan, kept only when the whole headline configuration measures faster twice; then an A/B through bench.py
out=gpurun_out/r6j; mkdir -p $out
cd $GRAFT_REPO_ROOT
( time timeout 2400 python tools/refine_plan.py --base demon_amd/tuned/plan_192x256_n32_l4.json --alt gpurun_in/alt_l4_r6g.json --alt demon_amd/tuned/plan_192x256_n32.json --alt gpurun_in/alt_l4_r6c.json --out $out/plan_192x256_n32_l4.json ) > $out/refine.log 2>&1
tail -25 $out/refine.log
q() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), round(d['value_single_lane'],1), d['config']['lanes'])"; }
B="python bench.py --no-cpu-baseline --no-e2e --no-roofline --steps 60 --warmup 5"
cp demon_amd/tuned/plan_192x256_n32_l4.json $out/plan_192x256_n32_l4_before.json
for rep in 1 2 3; do
  cp $out/plan_192x256_n32_l4_before.json demon_amd/tuned/plan_192x256_n32_l4.json
  timeout 300 $B 2>/dev/null | q "shipped plan" >> $out/ab.txt
  cp $out/plan_192x256_n32_l4.json demon_amd/tuned/plan_192x256_n32_l4.json
  timeout 300 $B 2>/dev/null | q "refined plan" >> $out/ab.txt
done
cat $out/ab.txt
