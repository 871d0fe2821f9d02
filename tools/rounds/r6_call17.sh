#!/bin/bash
# round 6, GPU call 17: wino_deconv variants 7 / 8 (32 channels x 48 / 64 tiles, one wave per SIMD, accumulators in AGPRs): tests (plain + poison),
# stand-alone probe of the transposed convs at batch 32, in-pass tune of the latency plan (transposed-conv candidates only), refinement of the
# throughput-mode plan with a forced-variant alternative, A/B
out=gpurun_out/r6r; mkdir -p $out
cd $GRAFT_REPO_ROOT
( time timeout 900 python -m pytest tests/test_variants_gpu.py tests/test_poison_gpu.py -q -p no:cacheprovider -k "deconv" ) > $out/tests.log 2>&1; echo "rc $?" >> $out/tests.log
tail -5 $out/tests.log
probe() { lab=$1; shift; echo "== $lab" >> $out/probe.txt; timeout 300 python tools/plan_probe.py deconv 32 $@ 8,0,1 8,1,1 8,2,1 8,3,1 8,4,1 8,5,1 8,7,1 8,8,1 8,7,2 8,5,2 2>/dev/null >> $out/probe.txt; }
probe "refine4 512->256 6x8"     512 6 8 256 4 4 2 2
probe "refine3 514->128 12x16"   514 12 16 128 4 4 2 2
probe "refine2 258->64 24x32"    258 24 32 64 4 4 2 2
probe "rf refine1 128->64 48x64" 128 48 64 64 4 4 2 2
probe "rf refine0 128->32 96x128" 128 96 128 32 4 4 2 2
cat $out/probe.txt
( time timeout 600 python tools/inpass_tune.py --batch 32 --kinds 8 --only upconv --rounds 1 --repeats 4 --out $out/plan_192x256_n32.json --verify 40 ) > $out/inpass_l1.log 2>&1
tail -12 $out/inpass_l1.log
# alternative for the throughput-mode plan: every big-map transposed conv on variant 7
python - <<'P'
import json
d = json.load(open("demon_amd/tuned/plan_192x256_n32_l4.json"))
for v, name in ((7, "alt7"), (8, "alt8")):
    p = dict(d["plan"])
    for k in p:
        if k.endswith("/upconv") and p[k][0] == 8 and "upsample" not in k:
            p[k] = [8, v, 1]
    json.dump(dict(d, plan=p), open("gpurun_out/r6r/%s.json" % name, "w"))
P
( time timeout 1200 python tools/refine_plan.py --base demon_amd/tuned/plan_192x256_n32_l4.json --alt $out/alt7.json --alt $out/alt8.json --out $out/plan_192x256_n32_l4.json ) > $out/refine.log 2>&1
grep -E "KEPT|refined plan|base plan|upconv" $out/refine.log
