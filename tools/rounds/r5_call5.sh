#!/bin/bash
# round 5, GPU call 5: stream priorities per lane (experiment), XCD order under lanes, throughput-mode plans tuned with FOUR passes in
# flight for the other shipped batch sizes / the v2 model
out=gpurun_out/r5e; mkdir -p $out
cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --no-e2e --no-roofline --steps 40 --warmup 5"
q() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), round(d['value_single_lane'],1), d['config']['lanes'], d['config']['lanes_mapping'])"; }
timeout 300 $B 2>/dev/null | q "base(l4 plan, q8)" >> $out/ab.txt
DEMON_LANE_PRIORITIES=1 timeout 300 $B 2>/dev/null | q "priorities hi/def/lo" >> $out/ab.txt
DEMON_LANE_PRIORITIES=2 timeout 300 $B 2>/dev/null | q "priorities leader+fillers" >> $out/ab.txt
DEMON_XCD_ORDER=0 timeout 300 $B 2>/dev/null | q "xcd order off" >> $out/ab.txt
timeout 300 $B 2>/dev/null | q "base again" >> $out/ab.txt
for n in 8 1 64; do
  ( time timeout 900 python tools/tune.py --batch $n --lanes 4 --rounds 3 --outdir $out ) > $out/tune_n${n}_l4.log 2>&1
done
( time timeout 900 python tools/tune.py --batch 32 --lanes 4 --rounds 3 --version 2 --outdir $out ) > $out/tune_v2_l4.log 2>&1
cp $out/plan_*_l4.json demon_amd/tuned/
for spec in "bootstrap:--workload bootstrap" "batch1:--batch 1" "batch8:--batch 8" "batch64:--batch 64" "v2:--workload v2"; do
  name=${spec%%:*}; args=${spec#*:}
  timeout 400 $B $args 2>/dev/null | q "$name/l4" >> $out/ab.txt
done
cat $out/ab.txt; tail -3 $out/tune_*_l4.log
