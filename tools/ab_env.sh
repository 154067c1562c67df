# (every arm under `timeout 180`: a runtime flag that hangs the process must not eat the session)
# usage: tools/ab_env.sh "<label>:<ENV=val ...>" ...   -- bench.py --no-extras per arm, one gpurun session
for arm in "$@"; do
  label=${arm%%:*}; envs=${arm#*:}
  env $envs timeout 180 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$label', d['ms_per_step'], d['value'])"
done
