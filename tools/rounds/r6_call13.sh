#!/bin/bash
# round 6, GPU call 13: the in-pass tuner (tools/inpass_tune.py) on the latency plan and on the throughput-mode plan of the metric's configuration
out=gpurun_out/r6n; mkdir -p $out
cd $GRAFT_REPO_ROOT
( time timeout 1500 python tools/inpass_tune.py --batch 32 --lanes 1 --rounds 2 --out $out/plan_192x256_n32.json --verify 40 ) > $out/inpass_l1.log 2>&1
tail -60 $out/inpass_l1.log
( time timeout 1800 python tools/inpass_tune.py --batch 32 --lanes 4 --rounds 2 --out $out/plan_192x256_n32_l4.json --verify 80 ) > $out/inpass_l4.log 2>&1
tail -60 $out/inpass_l4.log
q() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), round(d['value_single_lane'],1), d['config']['lanes'])"; }
B="python bench.py --no-cpu-baseline --no-e2e --no-roofline --steps 60 --warmup 5"
cp demon_amd/tuned/plan_192x256_n32.json $out/plan_192x256_n32_before.json
cp demon_amd/tuned/plan_192x256_n32_l4.json $out/plan_192x256_n32_l4_before.json
for rep in 1 2 3; do
  cp $out/plan_192x256_n32_before.json demon_amd/tuned/plan_192x256_n32.json; cp $out/plan_192x256_n32_l4_before.json demon_amd/tuned/plan_192x256_n32_l4.json
  timeout 300 $B 2>/dev/null | q "shipped plans" >> $out/ab.txt
  [ -s $out/plan_192x256_n32.json ] && cp $out/plan_192x256_n32.json demon_amd/tuned/plan_192x256_n32.json
  [ -s $out/plan_192x256_n32_l4.json ] && cp $out/plan_192x256_n32_l4.json demon_amd/tuned/plan_192x256_n32_l4.json
  timeout 300 $B 2>/dev/null | q "in-pass plans" >> $out/ab.txt
done
cat $out/ab.txt
