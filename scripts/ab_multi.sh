#!/bin/bash
# (the PETR_* switches only act on a tuning build: scripts/ab_build.sh tune -DPETR_TUNING_ENV; export PETR_HIP_LIB=.../libpetr_hip_tune.so)
# same-box A/B of several "VAR=value" settings (each given as one quoted argument, may hold several assignments),
# interleaved over 3 rounds, medians printed: scripts/ab_multi.sh "wl:dtype ..." "A=1" "A=0" "A=1 B=0" ...
CFGS=$1; shift
for cfg in $CFGS; do
  wl=${cfg%%:*}; dt=${cfg##*:}
  for round in 1 2 3; do
    i=0
    for setting in "$@"; do
      v=$(env $setting python bench.py --workload $wl --dtype $dt --steps 60 --warmup 10 --timed-only 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
      echo "$wl $dt [$setting] $v"
      i=$((i+1))
    done
  done
done | python -c "
import sys, collections, statistics
d = collections.OrderedDict()
for l in sys.stdin:
    k, v = l.rsplit(' ', 1)
    d.setdefault(k, []).append(float(v))
for k, v in d.items():
    print(f'{k}: median {statistics.median(v):.4f}  ({\" \".join(f\"{x:.3f}\" for x in v)})')
"
