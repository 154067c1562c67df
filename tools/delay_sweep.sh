# A/B in one gpurun session: does the chain hide behind the bulk when the bulk leaves room for it on every CU?
run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$1', d['ms_per_step'], d['value'])"; }
run base
OG_TILED_LDS_EXTRA=16384 run lds96
OG_TILED_LDS_EXTRA=16384 OG_ENGINE_BRANCH_DELAY=2 run lds96_delay2
OG_TILED_LDS_EXTRA=16384 OG_ENGINE_BRANCH_DELAY=3 run lds96_delay3
OG_TILED_LDS_EXTRA=16384 OG_ENGINE_BRANCH_DELAY=1 run lds96_delay1
