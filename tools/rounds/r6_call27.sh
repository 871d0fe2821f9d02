#!/bin/bash
# round 6, GPU call 27: under torch.distributed.run -- 8 / 12 / 16 hardware queues with the plateau calibration (round 5 chose 16 under a launcher with the
# full sweep + re-apply; does that still hold?)
out=gpurun_out/r6q2; mkdir -p $out
cd $GRAFT_REPO_ROOT
q() { python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); m=d['config']['lanes_mapping']; print('$1', round(d['value'],1), round(d['value_single_lane'],1), d['config']['lanes'], m['placeholder_streams'], m['attempts'], d['config']['hw_queues']['env'], {k:v for k,v in d['config']['lanes_calibration_pairs_per_s'].items() if k.startswith('4@') or k.startswith('3@')})"; }
port=29600
for rep in 1 2; do
  for hq in 8 12 16; do
    port=$((port+1))
    DEMON_HW_QUEUES=$hq DEMON_FORCE_DIST=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $port bench.py --gpus 1 --no-cpu-baseline --no-roofline --no-e2e --no-hoisted-leg 2>/dev/null | q "launcher hq=$hq" >> $out/hq.txt
  done
done
cat $out/hq.txt
