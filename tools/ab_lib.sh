# usage: tools/ab_lib.sh <tag|-> ...   -- bench.py --no-extras per variant library tools/build/libog_<tag>.so ("-" = the product library),
# one gpurun session; lines are diagnostic (a foreign library is loaded).  Every arm under `timeout 180`.
for tag in "$@"; do
  if [ "$tag" = "-" ]; then lib=""; else lib="OG_DECODER_LIB=$PWD/tools/build/libog_$tag.so"; fi
  env $lib timeout 180 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras --allow-diagnostic 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$tag', d['ms_per_step'], d['value'])"
done
