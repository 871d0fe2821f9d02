#!/bin/bash
# round 6, GPU call 7: both plans of the metric's configuration re-tuned on the round's kernels (five rounds each, majority vote), A/B against the
# shipped ones on this box; the plans are only shipped when they win both alternations
out=gpurun_out/r6g; mkdir -p $out
cd $GRAFT_REPO_ROOT
q() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), round(d['value_single_lane'],1), d['config']['lanes'])"; }
B="python bench.py --no-cpu-baseline --no-e2e --no-roofline --steps 60 --warmup 5"
( time timeout 1500 python tools/tune.py --batch 32 --lanes 4 --rounds 5 --outdir $out ) > $out/tune_l4.log 2>&1
( time timeout 1500 python tools/tune.py --batch 32 --lanes 1 --rounds 5 --outdir $out ) > $out/tune_l1.log 2>&1
cp demon_amd/tuned/plan_192x256_n32.json $out/plan_192x256_n32_before.json; cp demon_amd/tuned/plan_192x256_n32_l4.json $out/plan_192x256_n32_l4_before.json
for rep in 1 2 3; do
  cp $out/plan_192x256_n32_before.json demon_amd/tuned/plan_192x256_n32.json; cp $out/plan_192x256_n32_l4_before.json demon_amd/tuned/plan_192x256_n32_l4.json
  timeout 300 $B 2>/dev/null | q "shipped plans" >> $out/ab.txt
  cp $out/plan_192x256_n32.json $out/plan_192x256_n32_l4.json demon_amd/tuned/
  timeout 300 $B 2>/dev/null | q "re-tuned plans" >> $out/ab.txt
done
cat $out/ab.txt; tail -4 $out/tune_l4.log $out/tune_l1.log
python - <<'PY'
import json
for f in ("plan_192x256_n32.json", "plan_192x256_n32_l4.json"):
    a = json.load(open("gpurun_out/r6g/" + f.replace(".json", "_before.json")))["plan"]; b = json.load(open("gpurun_out/r6g/" + f))["plan"]
    ch = {k: (a[k], b[k]) for k in b if a.get(k) != b[k]}
    print(f, len(ch), "layers changed"); [print("  ", k, v) for k, v in sorted(ch.items())]
PY
