#!/bin/bash
# round 6, GPU call 16: the throughput-mode plans of the other workloads refined on their own metric (tools/refine_plan.py), alternatives = the
# in-pass latency plan of the same workload; the metric's plan once more with the in-pass-under-lanes plan as a further alternative
out=gpurun_out/r6q; mkdir -p $out
cd $GRAFT_REPO_ROOT
T=demon_amd/tuned
ref() {  # name lanes steps alts...
  name=$1; lanes=$2; steps=$3; shift 3
  alts=""; for a in "$@"; do alts="$alts --alt $a"; done
  ( time timeout 1500 python tools/refine_plan.py --base $T/$name.json $alts --lanes $lanes --steps $steps --out $out/$name.json ) > $out/refine_$name.log 2>&1
  grep -E "KEPT|refined plan|base plan" $out/refine_$name.log
}
ref plan_192x256_n32_l4 4 48 gpurun_in/alt_l4_r6n.json gpurun_in/alt_l4_r6c.json
ref plan_v2_192x256_n32_l4 4 48 $T/plan_v2_192x256_n32.json
ref plan_192x256_n64_l4 4 24 $T/plan_192x256_n64.json
ref plan_480x640_n64_l2 2 6 $T/plan_480x640_n64.json
ref plan_192x256_n8_l4 4 160 $T/plan_192x256_n8.json
ref plan_192x256_n1_l4 4 600 $T/plan_192x256_n1.json
ls -la $out
