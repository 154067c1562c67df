# usage: tools/ab_k1_lib.sh "<label>:<ENV=val ...>" ...  -- K1 inside the driver's timed region (roofline.us_per_launch) per arm;
# an arm may name another library build with OG_DECODER_LIB=tools/build/libog_<tag>.so (bench.py then runs with --allow-diagnostic)
for arm in "$@"; do
  label=${arm%%:*}; envs=${arm#*:}
  flag=; case "$envs" in *OG_DECODER_LIB*) flag=--allow-diagnostic;; esac
  env $envs timeout 180 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras $flag 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); r=d['roofline']; print('$label', 'K1 us', r['us_per_launch'], 'frac', r['frac'], 'ms/step', d['ms_per_step'], 'k3', d.get('stage_us', {}).get('k3_group'))"
done
