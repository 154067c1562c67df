mkdir -p gpurun_out/r03i
for w in none chain 1x1 s2big c160 c80 c40; do
  if [ $w = none ]; then unset OG_ENGINE_WHATIF; else export OG_ENGINE_WHATIF=$w; fi
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --allow-diagnostic 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$w', d['ms_per_step'], d['value'])"
done
unset OG_ENGINE_WHATIF
OG_CONV_TILED=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('tiled=0', d['ms_per_step'], d['value'])"
OG_CONV_TILED=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('tiled=1', d['ms_per_step'], d['value'])"
