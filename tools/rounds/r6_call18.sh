#!/bin/bash
# round 6, GPU call 18: conv1 / conv2 of the depth+motion blocks hoisted onto the side stream behind the flow block's conv3_1: exactness tests,
# then one lane with the option on / off (graph replays alternating in one process), batch 32 / 8 / 1 and the other workloads through bench.py
out=gpurun_out/r6s; mkdir -p $out
cd $GRAFT_REPO_ROOT
( time timeout 900 python -m pytest tests/test_nets_gpu.py -q -p no:cacheprovider -k "hoisted or side_branches or reuse_image" ) > $out/tests.log 2>&1; echo "rc $?" >> $out/tests.log
tail -5 $out/tests.log
python - > $out/ab.txt 2>&1 <<'P'
import sys, time, numpy as np
sys.path.insert(0, ".")
import demon_amd.lanes
from demon_amd import DemonContext, weights as W
for (n, H, Wd, ver, steps) in ((32, 192, 256, 1, 40), (8, 192, 256, 1, 100), (1, 192, 256, 1, 300), (64, 192, 256, 1, 20), (32, 192, 256, 2, 40), (64, 480, 640, 1, 5)):
    ctx = DemonContext(0, n, H, Wd, version=ver)
    ctx.set_weights(W.synthetic_weights(seed=1, height=H, width=Wd, version=ver))
    ctx.load_tuned_plan(n)
    rng = np.random.default_rng(0)
    pair = rng.random((n, 6, H, Wd), dtype=np.float32) - np.float32(0.5)
    ctx.upload_inputs(pair, pair[:, 3:6].reshape(n, 3, H // 4, 4, Wd // 4, 4).mean(axis=(3, 5)).astype(np.float32))
    res = {0: [], 1: []}
    for rep in range(3):
        for hoist in (0, 1):
            ctx.set_option("hoist_image_layers", hoist)
            for _ in range(3): ctx.run_full(n, 3)
            ctx.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps): ctx.run_full(n, 3)
            ctx.synchronize()
            res[hoist].append(n * steps / (time.perf_counter() - t0))
    print("batch %d %dx%d v%d: at their own place %s | hoisted %s pairs/s" % (n, H, Wd, ver, " ".join("%.1f" % v for v in res[0]), " ".join("%.1f" % v for v in res[1])), flush=True)
    ctx.close()
P
cat $out/ab.txt
timeout 400 python bench.py --no-cpu-baseline --no-e2e --no-roofline > $out/bench.json 2> $out/bench.err
python -c "import json; d=json.load(open('$out/bench.json')); print(d['value'], d['value_single_lane'], d.get('value_image_features_hoisted'))"
