#!/bin/bash
# round 5, last GPU call: the bench records, the RCCL single-rank record and the -rs test log again on the final library (after the
# stream-exchange / tuner-stream fixes; kernels and plans unchanged: the rocprofv3 / PMC summaries of collect_profiles.sh stay valid),
# plus the run-to-run spread of the headline
tag=r05
out=gpurun_out/${tag}_profiles; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 500 python bench.py --layers > $out/bench.json 2> $out/layers_hipevents.txt
for spec in "config1_bootstrap:--workload bootstrap" "batch1:--batch 1" "batch8:--batch 8" "batch64:--batch 64" "config4_hires:--workload hires --layers" "v2:--workload v2 --layers"; do
  name=${spec%%:*}; args=${spec#*:}
  timeout 400 python bench.py $args --no-cpu-baseline > $out/bench_$name.json 2> $out/layers_$name.txt
done
for i in 1 2 3 4 5; do
  timeout 300 python bench.py --no-cpu-baseline --no-e2e --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['value_single_lane'],1), d['config']['lanes'], d['config']['lanes_mapping']['placeholder_streams'])" >> $out/spread.txt
done
DEMON_FORCE_DIST=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 1 --no-cpu-baseline --no-roofline --no-e2e > $out/forcedist.out 2> $out/forcedist.err
timeout 1500 python -m pytest tests -m gpu -q -rs -p no:cacheprovider > $out/gputest_rs.log 2>&1; echo "pytest rc $?" >> $out/gputest_rs.log
cat $out/spread.txt; tail -3 $out/gputest_rs.log
