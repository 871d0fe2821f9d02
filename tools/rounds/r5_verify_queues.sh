q() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), round(d['value_single_lane'],1), d['config']['lanes'], d['config']['lanes_mapping'], d['config']['plan_setup_s'])"; }
B="bench.py --no-cpu-baseline --no-roofline --no-e2e"
python $B 2>/dev/null | q "plain"
for i in 1 2 3; do DEMON_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29577 $B --gpus 1 2>/dev/null | grep "^{" | q "torchrun"; done
python $B 2>/dev/null | q "plain"
python -m pytest tests/test_nets_gpu.py tests/test_bench_gpu.py tests/test_fullsize_gpu.py -q -x -k "lane or borrowed or released or bench" -p no:cacheprovider 2>&1 | tail -3
