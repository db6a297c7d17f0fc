#!/bin/bash
# same-box A/B of the fp32 attention backward forms (round 3): the per-(key block, head, query split) launch of round 2
# (PETR_MHA_BWD_SK=0) against the persistent form (1), with one or two resident workgroups per CU
cd $GRAFT_REPO_ROOT
for p in 0.0 0.1; do
  for cfg in "PETR_MHA_BWD_SK=0" "PETR_MHA_BWD_SK=1 PETR_MHA_BWD_SK_SLOTS=2" "PETR_MHA_BWD_SK=1 PETR_MHA_BWD_SK_SLOTS=1"; do
    echo "== drop $p  $cfg"
    env $cfg python scripts/bwd32_time.py $p 2>&1 | grep -v amdgpu.ids
  done
done
