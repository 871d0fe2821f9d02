#!/bin/bash
# round 5, GPU call 4: 8 hardware queues as the package default -- every workload against 4 queues; side branches inside a lane
# group; a launch plan tuned with FOUR passes in flight
out=gpurun_out/r5d; mkdir -p $out
cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --no-e2e --no-roofline --steps 40 --warmup 5"
q() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), round(d['value_single_lane'],1), d['config']['lanes'], d['config']['lanes_mapping'])"; }
timeout 300 $B 2>/dev/null | q "default(8)" >> $out/ab.txt
for spec in "full:" "bootstrap:--workload bootstrap" "batch1:--batch 1" "batch8:--batch 8" "batch64:--batch 64" "hires:--workload hires --steps 10" "v2:--workload v2"; do
  name=${spec%%:*}; args=${spec#*:}
  GPU_MAX_HW_QUEUES=4 timeout 400 $B $args 2>/dev/null | q "$name/q4" >> $out/ab.txt
  timeout 400 $B $args 2>/dev/null | q "$name/q8" >> $out/ab.txt
done
DEMON_LANES_SIDE_BRANCHES=1 timeout 300 $B 2>/dev/null | q "side_branches_on/q8" >> $out/ab.txt
( time timeout 900 python tools/tune.py --batch 32 --lanes 4 --rounds 3 --outdir $out ) > $out/tune_l4.log 2>&1
cp $out/plan_192x256_n32_l4.json demon_amd/tuned/ 2>/dev/null
timeout 300 $B 2>/dev/null | q "plan_l4/q8" >> $out/ab.txt
timeout 300 $B 2>/dev/null | q "plan_l4/q8" >> $out/ab.txt
cat $out/ab.txt; tail -4 $out/tune_l4.log
