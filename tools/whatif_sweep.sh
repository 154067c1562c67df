# the step with one class of layers removed (OG_ENGINE_WHATIF, wrong results): what each class costs in the network
for w in none chain 1x1 s2big c160 c80 c40; do
  if [ $w = none ]; then unset OG_ENGINE_WHATIF; else export OG_ENGINE_WHATIF=$w; fi
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --allow-diagnostic 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$w', d['ms_per_step'], d['value'])"
done
unset OG_ENGINE_WHATIF
