#!/bin/bash
# round 6, GPU call 5: branch-free packed epilogues + the in-workgroup K split of the small-map transposed conv (wino_deconv variant 6): every
# variant test (no -x: the list of failures matters), the refine4 probe, refine4 re-tuned in both plans, A/B
out=gpurun_out/r6e; mkdir -p $out
cd $GRAFT_REPO_ROOT
( time timeout 1500 python -m pytest tests/test_variants_gpu.py tests/test_layers_gpu.py -q -p no:cacheprovider ) > $out/tests.log 2>&1
tail -8 $out/tests.log
timeout 300 python tools/refine4_probe.py > $out/refine4_probe.txt 2>&1; cat $out/refine4_probe.txt
q() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), round(d['value_single_lane'],1), d['config']['lanes'])"; }
B="python bench.py --no-cpu-baseline --no-e2e --no-roofline --steps 60 --warmup 5"
timeout 300 $B 2>/dev/null | q "shipped plans" >> $out/ab.txt
( time timeout 600 python tools/tune.py --batch 32 --lanes 1 --rounds 3 --outdir $out --only refine4 ) > $out/tune_l1.log 2>&1
( time timeout 600 python tools/tune.py --batch 32 --lanes 4 --rounds 3 --outdir $out --only refine4 ) > $out/tune_l4.log 2>&1
cp demon_amd/tuned/plan_192x256_n32.json $out/plan_192x256_n32_before.json; cp demon_amd/tuned/plan_192x256_n32_l4.json $out/plan_192x256_n32_l4_before.json
cp $out/plan_192x256_n32.json $out/plan_192x256_n32_l4.json demon_amd/tuned/
timeout 300 $B 2>/dev/null | q "refine4 re-tuned" >> $out/ab.txt
timeout 300 $B 2>/dev/null | q "refine4 re-tuned" >> $out/ab.txt
cp $out/plan_192x256_n32_before.json demon_amd/tuned/plan_192x256_n32.json; cp $out/plan_192x256_n32_l4_before.json demon_amd/tuned/plan_192x256_n32_l4.json
timeout 300 $B 2>/dev/null | q "shipped plans" >> $out/ab.txt
cat $out/ab.txt
python - <<'PY'
import json
for f in ("plan_192x256_n32.json", "plan_192x256_n32_l4.json"):
    a = json.load(open("gpurun_out/r6e/" + f.replace(".json", "_before.json")))["plan"]; b = json.load(open("gpurun_out/r6e/" + f))["plan"]
    ch = {k: (a[k], b[k]) for k in b if a.get(k) != b[k]}
    print(f, len(ch), "layers changed"); [print("  ", k, v) for k, v in sorted(ch.items())]
PY
