#!/bin/bash
# round 6, GPU call 19: the metric's throughput-mode plan refined once more, alternatives = the three-lane plan, the batch-64 and batch-16 throughput plans
out=gpurun_out/r6t; mkdir -p $out
cd $GRAFT_REPO_ROOT
T=demon_amd/tuned
( time timeout 1500 python tools/refine_plan.py --base $T/plan_192x256_n32_l4.json --alt $T/plan_192x256_n32_l3.json --alt $T/plan_192x256_n64_l4.json --alt $T/plan_192x256_n16_l3.json --alt $T/plan_192x256_n64.json --lanes 4 --steps 48 --out $out/plan_192x256_n32_l4.json ) > $out/refine.log 2>&1
grep -E "KEPT|refined plan|base plan" $out/refine.log
( time timeout 1500 python tools/refine_plan.py --base $T/plan_v2_192x256_n32_l4.json --alt $T/plan_v2_192x256_n32_l3.json --lanes 4 --steps 48 --out $out/plan_v2_192x256_n32_l4.json ) > $out/refine_v2.log 2>&1
grep -E "KEPT|refined plan|base plan" $out/refine_v2.log
