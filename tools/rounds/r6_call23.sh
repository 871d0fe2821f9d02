#!/bin/bash
# round 6, GPU call 23: the lane calibration that stops on a plateau: bench.py under torch.distributed.run (RCCL route, 16 hardware queues) three times,
# plain twice; then the driver's own commands -- the -m gpu suite (pytest.ini / conftest as shipped) and smoke()
out=gpurun_out/r6x; mkdir -p $out
cd $GRAFT_REPO_ROOT
q() { python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); m=d['config']['lanes_mapping']; print('$1', round(d['value'],1), round(d['value_single_lane'],1), d['config']['lanes'], {k:m[k] for k in ('placeholder_streams','pairs_per_s','verified_pairs_per_s','attempts','reproduced')}, len(d['config']['lanes_calibration_pairs_per_s']), 'cells')"; }
for i in 1 2 3; do
  DEMON_FORCE_DIST=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2957$i bench.py --gpus 1 --no-cpu-baseline --no-roofline --no-e2e > $out/forcedist$i.out 2> $out/forcedist$i.err
  cat $out/forcedist$i.out | q "launcher" >> $out/calib.txt
done
for i in 1 2; do
  timeout 400 python bench.py --no-cpu-baseline --no-roofline --no-e2e > $out/plain$i.out 2> $out/plain$i.err
  cat $out/plain$i.out | q "plain" >> $out/calib.txt
done
cat $out/calib.txt
grep "^{" $out/forcedist3.out | tail -1 > profiles/r06_forcedist_rccl_1rank.json; cp profiles/r06_forcedist_rccl_1rank.json $out/
( time timeout 1500 python -m pytest tests/ -x -q -m gpu ) > $out/gputest.log 2>&1; echo "pytest rc $?" >> $out/gputest.log
tail -5 $out/gputest.log
( time timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > $out/smoke.log 2>&1; tail -3 $out/smoke.log
