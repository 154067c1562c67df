run() { python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras $2 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$1', d['ms_per_step'], d['value'])"; }
run base ""
run overlap "--overlap"
run base2 ""
run overlap2 "--overlap"
