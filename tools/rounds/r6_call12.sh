#!/bin/bash
# round 6, GPU call 12: conv_wino4 flat line order (plan field ksplit = 3): tests (plain + poison harness), stand-alone probe of the level-4 / 5
# layers at batch 32 (per-image tiles vs flat, every shape), then a re-tune of those layers of both plans and an A/B through bench.py
out=gpurun_out/r6m; mkdir -p $out
cd $GRAFT_REPO_ROOT
( time timeout 900 python -m pytest tests/test_variants_gpu.py tests/test_poison_gpu.py -q -p no:cacheprovider -k "four_outputs or flat_line" ) > $out/tests.log 2>&1; echo "rc $?" >> $out/tests.log
tail -6 $out/tests.log
probe() {  # label cin h w cout kh kw sh sw current-plans...
  lab=$1; shift
  plans=""
  for v in 0 1 2 3 4 5 6 7 8 9 10 11 12 13; do plans="$plans 16,$v,1 16,$v,3"; done
  echo "== $lab" >> $out/probe.txt
  timeout 300 python tools/plan_probe.py conv 32 $@ $plans 2>/dev/null | grep "wino4<" | sort -u -k6,6 >> $out/probe.txt
}
probe "conv4y 128->256 5x1 s2 24x32"  128 24 32 256 5 1 2 1
probe "conv4x 256->256 1x5 s2 12x32"  256 12 32 256 1 5 1 2
probe "conv4_1y 256->256 3x1 12x16"   256 12 16 256 3 1 1 1
probe "conv4_1x 256->256 1x3 12x16"   256 12 16 256 1 3 1 1
probe "conv5y 256->512 5x1 s2 12x16"  256 12 16 512 5 1 2 1
probe "conv5x 512->512 1x5 s2 6x16"   512 6 16 512 1 5 1 2
probe "conv5_1y 512->512 3x1 6x8"     512 6 8 512 3 1 1 1
probe "conv5_1x 512->512 1x3 6x8"     512 6 8 512 1 3 1 1
cat $out/probe.txt
# re-tune conv4* / conv5* of both plans with the flat forms in the candidate list, then A/B
cp demon_amd/tuned/plan_192x256_n32.json $out/plan_192x256_n32_before.json
cp demon_amd/tuned/plan_192x256_n32_l4.json $out/plan_192x256_n32_l4_before.json
( time timeout 900 python tools/tune.py --batch 32 --rounds 3 --only conv4,conv5 --outdir $out ) > $out/tune_l1.log 2>&1
( time timeout 1200 python tools/tune.py --batch 32 --rounds 3 --lanes 4 --only conv4,conv5 --outdir $out ) > $out/tune_l4.log 2>&1
tail -3 $out/tune_l1.log $out/tune_l4.log
python - <<'P' >> $out/plan_diff.txt
import json
for f in ("plan_192x256_n32", "plan_192x256_n32_l4"):
    a = json.load(open("gpurun_out/r6m/%s_before.json" % f))["plan"]; b = json.load(open("gpurun_out/r6m/%s.json" % f))["plan"]
    for k in sorted(a):
        if a[k] != b.get(k): print(f, k, a[k], "->", b.get(k))
P
cat $out/plan_diff.txt
q() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), round(d['value_single_lane'],1), d['config']['lanes'])"; }
B="python bench.py --no-cpu-baseline --no-e2e --no-roofline --steps 60 --warmup 5"
for rep in 1 2 3; do
  cp $out/plan_192x256_n32_before.json demon_amd/tuned/plan_192x256_n32.json; cp $out/plan_192x256_n32_l4_before.json demon_amd/tuned/plan_192x256_n32_l4.json
  timeout 300 $B 2>/dev/null | q "shipped plans" >> $out/ab.txt
  cp $out/plan_192x256_n32.json demon_amd/tuned/plan_192x256_n32.json; cp $out/plan_192x256_n32_l4.json demon_amd/tuned/plan_192x256_n32_l4.json
  timeout 300 $B 2>/dev/null | q "re-tuned (flat forms)" >> $out/ab.txt
done
cat $out/ab.txt
