#!/bin/bash
# round 6, GPU call 1: the new parity tests of the headline's protocol (plain / under torch.distributed.run / two processes at once),
# the lane-mode A/B (calibrated round robin vs CU-mask partitions vs one graph per lane group), NUMA binding A/B, and the per-launch
# table of the LATENCY plan (one pass at a time: the regime of value_single_lane)
out=gpurun_out/r6a; mkdir -p $out
cd $GRAFT_REPO_ROOT
( time timeout 1700 python -m pytest tests/test_fullsize_gpu.py::test_config2_headline_protocol_every_pair_of_every_kept_lane tests/test_bench_gpu.py::test_headline_protocol_parity_under_torch_distributed_run tests/test_bench_gpu.py::test_two_processes_calibrate_their_lanes_concurrently_on_one_gpu -x -q -p no:cacheprovider ) > $out/tests.log 2>&1
tail -5 $out/tests.log
timeout 200 python tools/lane_parity.py --lanes 5 > $out/lane_parity.json 2> $out/lane_parity.err
( time timeout 1200 python tools/lane_modes.py --steps 40 ) > $out/lane_modes.txt 2> $out/lane_modes.err
cat $out/lane_modes.txt
timeout 300 python bench.py --lanes 1 --layers --no-cpu-baseline --no-e2e > $out/bench_lat.json 2> $out/layers_lat.txt
q() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), round(d['value_single_lane'],1), d['config']['lanes'], d['config'].get('lanes_mapping'), d['per_rank']['ranks'][0].get('host_placement'))"; }
B="python bench.py --no-cpu-baseline --no-e2e --no-roofline --steps 40 --warmup 5"
DEMON_BIND_NUMA=0 timeout 300 $B 2>/dev/null | q "unbound" >> $out/ab.txt
DEMON_BIND_NUMA=1 timeout 300 $B 2>/dev/null | q "bound" >> $out/ab.txt
DEMON_BIND_NUMA=0 timeout 300 $B 2>/dev/null | q "unbound" >> $out/ab.txt
DEMON_BIND_NUMA=1 timeout 300 $B 2>/dev/null | q "bound" >> $out/ab.txt
cat $out/ab.txt
( time timeout 400 python bench.py --steps 20 ) > $out/bench_default.json 2> $out/bench_default.err
tail -c 1500 $out/bench_default.json
