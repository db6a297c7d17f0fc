for round in 1 2 3; do
for setting in "PETR_X=0" "PETR_MHA_BWD_QSPLITS=5" "PETR_MHA_BWD_QSPLITS=3"; do
  v=$(env $setting python bench.py --workload c5 --dtype fp32 --batch 2 --steps 40 --warmup 8 --timed-only 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
  echo "B2 [$setting] $v"
done; done
