#!/bin/bash
# round 6, GPU call 4: packed-fp32 output transforms (wino4 / wino1d / wino3rows / conv_row epilogues on floatx4) -- every variant test, then the
# headline A/B against the previous build's library (gpurun_in/libdemon_hip_prev.so), with shader clock / power sampled during the runs
out=gpurun_out/r6d; mkdir -p $out
cd $GRAFT_REPO_ROOT
( time timeout 1500 python -m pytest tests/test_variants_gpu.py tests/test_layers_gpu.py -x -q -p no:cacheprovider ) > $out/tests.log 2>&1
tail -4 $out/tests.log
q() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), round(d['value_single_lane'],1), d['config']['lanes'])"; }
B="python bench.py --no-cpu-baseline --no-e2e --no-roofline --steps 60 --warmup 5"
smi() { for i in 1 2 3 4 5 6 7 8 9 10 11 12; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|mclk" | tr '\n' ' '; echo; sleep 1.5; done; }
for rep in 1 2; do
  DEMON_HIP_LIB=$PWD/gpurun_in/libdemon_hip_prev.so timeout 300 $B 2>/dev/null | q "previous build" >> $out/ab.txt
  timeout 300 $B 2>/dev/null | q "packed epilogues" >> $out/ab.txt
done
( smi > $out/smi_during.txt ) &
timeout 300 python bench.py --no-cpu-baseline --no-e2e --no-roofline --steps 400 --warmup 5 2>/dev/null | q "long run (clock sampling)" >> $out/ab.txt
wait
smi | head -3 > $out/smi_idle.txt
cat $out/ab.txt; echo; head -12 $out/smi_during.txt; echo idle; cat $out/smi_idle.txt
timeout 300 python bench.py --lanes 1 --layers --no-cpu-baseline --no-e2e > $out/bench_lat.json 2> $out/layers_lat.txt
