q() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), d['config']['lanes'], round(d['extra']['end_to_end_ms_per_step'],2), d['extra'].get('pipelined',{}).get('pairs_per_s'))"; }
B="python bench.py --no-cpu-baseline --steps 20 --warmup 3"
$B 2>/dev/null | q "q8 auto roofline, fresh streams + tune streams dropped"
DEMON_BENCH_FRESH_STREAMS=0 $B 2>/dev/null | q "q8 auto roofline, tune streams dropped only"
