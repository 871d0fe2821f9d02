#!/bin/bash
# round 6, GPU call 2: (1) the new wino4 shapes (three lines per wave), the tile-walking form and the narrow small-head tiles against PyTorch;
# (2) lanes on CU partitions WITH a calibrated stream mapping (call 1: one lane on a quarter of the CUs runs 0.40 of the whole chip's rate,
# but four uncalibrated partition lanes shared hardware queues); (3) what the new wino4 forms are worth: the 3-tap layers re-tuned, A/B
out=gpurun_out/r6b; mkdir -p $out
cd $GRAFT_REPO_ROOT
( time timeout 900 python -m pytest tests/test_variants_gpu.py -x -q -p no:cacheprovider -k "four_outputs or tile_walking or small_heads" ) > $out/tests.log 2>&1
tail -4 $out/tests.log
M="rr,pmask-4-4c,pmask-4-4cl,pmask-4-2c,pmask-2-2c,pmask-3-1c,rr"
( time timeout 900 python tools/lane_modes.py --steps 40 --modes $M ) > $out/lane_modes.txt 2> $out/lane_modes.err
DEMON_HW_QUEUES=16 timeout 400 python tools/lane_modes.py --steps 48 --modes pmask-8-4c,pmask-6-2c,pmask-4-4c >> $out/lane_modes.txt 2>> $out/lane_modes.err
cat $out/lane_modes.txt
q() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), round(d['value_single_lane'],1), d['config']['lanes'])"; }
B="python bench.py --no-cpu-baseline --no-e2e --no-roofline --steps 40 --warmup 5"
timeout 300 $B 2>/dev/null | q "shipped plans" >> $out/ab.txt
( time timeout 900 python tools/tune.py --batch 32 --lanes 1 --rounds 3 --outdir $out --only _1,conv2_extra_inputsx ) > $out/tune_l1.log 2>&1
( time timeout 900 python tools/tune.py --batch 32 --lanes 4 --rounds 3 --outdir $out --only _1,conv2_extra_inputsx ) > $out/tune_l4.log 2>&1
cp demon_amd/tuned/plan_192x256_n32.json $out/plan_192x256_n32_before.json; cp demon_amd/tuned/plan_192x256_n32_l4.json $out/plan_192x256_n32_l4_before.json
cp $out/plan_192x256_n32.json $out/plan_192x256_n32_l4.json demon_amd/tuned/
timeout 300 $B 2>/dev/null | q "3-tap layers re-tuned" >> $out/ab.txt
timeout 300 $B 2>/dev/null | q "3-tap layers re-tuned" >> $out/ab.txt
cp $out/plan_192x256_n32_before.json demon_amd/tuned/plan_192x256_n32.json; cp $out/plan_192x256_n32_l4_before.json demon_amd/tuned/plan_192x256_n32_l4.json
timeout 300 $B 2>/dev/null | q "shipped plans" >> $out/ab.txt
cat $out/ab.txt
python - <<'PY'
import json
for f in ("plan_192x256_n32.json", "plan_192x256_n32_l4.json"):
    a = json.load(open("gpurun_out/r6b/" + f.replace(".json", "_before.json")))["plan"]; b = json.load(open("gpurun_out/r6b/" + f))["plan"]
    ch = {k: (a[k], b[k]) for k in b if a.get(k) != b[k]}
    print(f, len(ch), "layers changed"); [print("  ", k, v) for k, v in sorted(ch.items())]
PY
