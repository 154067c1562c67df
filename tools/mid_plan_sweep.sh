for plan in "" "128,3,3,64" "128,2,3,64" "128,6,3,64" "64,3,4" "128,3,3"; do
  echo "MID_PLAN=$plan"
  OG_CONV_MID_PLAN=$plan python tools/conv_bench.py --only 2 7 --reps 10 2>&1 | grep -v amdgpu | cut -c1-200
done
