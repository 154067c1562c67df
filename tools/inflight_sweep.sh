# hardware queues x branch forking, one gpurun session
run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras $2 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$1', d['ms_per_step'], d['value'])"; }
run q4 ""
GPU_MAX_HW_QUEUES=1 run q1 ""
GPU_MAX_HW_QUEUES=2 run q2 ""
GPU_MAX_HW_QUEUES=3 run q3 ""
OG_ENGINE_BRANCHES=0 run nobranch_q4 ""
OG_ENGINE_BRANCH_MAX_DEPTH=0 run branch_depth0 ""
OG_ENGINE_BRANCH_MAX_DEPTH=1 run branch_depth1 ""
OG_ENGINE_BRANCH_MAX_DEPTH=2 run branch_depth2 ""
GPU_MAX_HW_QUEUES=2 OG_ENGINE_BRANCH_MAX_DEPTH=2 run q2_depth2 ""
