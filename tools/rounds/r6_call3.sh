#!/bin/bash
# round 6, GPU call 3 (call 2 again after the shape-table fix): the new wino4 shapes / tile-walking form / narrow small-head tiles against
# PyTorch, then what they are worth: the 3-tap layers re-tuned in both plans, A/B on this box
out=gpurun_out/r6c; mkdir -p $out
cd $GRAFT_REPO_ROOT
( time timeout 1200 python -m pytest tests/test_variants_gpu.py -x -q -p no:cacheprovider -k "four_outputs or tile_walking or small_heads" ) > $out/tests.log 2>&1
tail -4 $out/tests.log
q() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), round(d['value_single_lane'],1), d['config']['lanes'])"; }
B="python bench.py --no-cpu-baseline --no-e2e --no-roofline --steps 40 --warmup 5"
timeout 300 $B 2>/dev/null | q "shipped plans" >> $out/ab.txt
( time timeout 900 python tools/tune.py --batch 32 --lanes 1 --rounds 3 --outdir $out --only _1,conv2_extra_inputsx ) > $out/tune_l1.log 2>&1
( time timeout 900 python tools/tune.py --batch 32 --lanes 4 --rounds 3 --outdir $out --only _1,conv2_extra_inputsx ) > $out/tune_l4.log 2>&1
cp demon_amd/tuned/plan_192x256_n32.json $out/plan_192x256_n32_before.json; cp demon_amd/tuned/plan_192x256_n32_l4.json $out/plan_192x256_n32_l4_before.json
cp $out/plan_192x256_n32.json $out/plan_192x256_n32_l4.json demon_amd/tuned/
timeout 300 $B 2>/dev/null | q "3-tap layers re-tuned" >> $out/ab.txt
timeout 300 $B 2>/dev/null | q "3-tap layers re-tuned" >> $out/ab.txt
timeout 300 python bench.py --lanes 1 --layers --no-cpu-baseline --no-e2e > $out/bench_lat.json 2> $out/layers_lat.txt
cp $out/plan_192x256_n32_before.json demon_amd/tuned/plan_192x256_n32.json; cp $out/plan_192x256_n32_l4_before.json demon_amd/tuned/plan_192x256_n32_l4.json
timeout 300 $B 2>/dev/null | q "shipped plans" >> $out/ab.txt
cat $out/ab.txt
python - <<'PY'
import json
for f in ("plan_192x256_n32.json", "plan_192x256_n32_l4.json"):
    a = json.load(open("gpurun_out/r6c/" + f.replace(".json", "_before.json")))["plan"]; b = json.load(open("gpurun_out/r6c/" + f))["plan"]
    ch = {k: (a[k], b[k]) for k in b if a.get(k) != b[k]}
    print(f, len(ch), "layers changed"); [print("  ", k, v) for k, v in sorted(ch.items())]
PY
