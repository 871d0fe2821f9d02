#!/bin/bash
# round 6, GPU call 26: the driver's own commands on the final tree, twice (statistics for the silent abort under the shipped pytest configuration)
out=gpurun_out/r6zz; mkdir -p $out
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  ( time timeout 1500 python -m pytest tests/ -x -q -m gpu ) > $out/gputest$i.log 2>&1; echo "pytest rc $?" >> $out/gputest$i.log
  tail -6 $out/gputest$i.log | grep -E "passed|failed|rc|Abort|real"
done
( time timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > $out/smoke.log 2>&1; tail -2 $out/smoke.log
( time timeout 600 python bench.py ) > $out/bench.json 2> $out/bench.err; python -c "
import json; d=json.loads([l for l in open('$out/bench.json') if l.startswith('{')][-1]); print(d['value'], d['value_single_lane'], d.get('value_image_features_hoisted'), d['config']['lanes_mapping']['attempts'], d['config']['plan_setup_s'], d['cpu_baseline']['value'])"
