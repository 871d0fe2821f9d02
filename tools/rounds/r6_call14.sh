#!/bin/bash
# round 6, GPU call 14: the in-pass tuner on the latency plans of the other batch sizes / models (batch 1: the reference's real caller), each with an
# alternating verification of its own; bench with the hoisted-features leg
out=gpurun_out/r6o; mkdir -p $out
cd $GRAFT_REPO_ROOT
( time timeout 900 python tools/inpass_tune.py --batch 1 --ks 1,2,3,4,6,8,12,16,24,32 --rounds 2 --repeats 5 --out $out/plan_192x256_n1.json --verify 200 ) > $out/inpass_n1.log 2>&1
tail -12 $out/inpass_n1.log
( time timeout 900 python tools/inpass_tune.py --batch 8 --ks 1,2,3,4,6,8,12,16 --rounds 2 --repeats 4 --out $out/plan_192x256_n8.json --verify 100 ) > $out/inpass_n8.log 2>&1
tail -8 $out/inpass_n8.log
( time timeout 900 python tools/inpass_tune.py --batch 64 --rounds 2 --out $out/plan_192x256_n64.json --verify 30 ) > $out/inpass_n64.log 2>&1
tail -8 $out/inpass_n64.log
( time timeout 900 python tools/inpass_tune.py --batch 32 --version 2 --rounds 2 --out $out/plan_v2_192x256_n32.json --verify 40 ) > $out/inpass_v2.log 2>&1
tail -8 $out/inpass_v2.log
( time timeout 1200 python tools/inpass_tune.py --batch 64 --height 480 --width 640 --rounds 1 --repeats 2 --out $out/plan_480x640_n64.json --verify 6 ) > $out/inpass_hires.log 2>&1
tail -8 $out/inpass_hires.log
timeout 400 python bench.py --no-cpu-baseline --no-e2e --no-roofline > $out/bench.json 2> $out/bench.err
python -c "import json; d=json.load(open('$out/bench.json')); print(d['value'], d['value_single_lane'], d.get('value_image_features_hoisted'), d.get('image_features_hoisted'))"
