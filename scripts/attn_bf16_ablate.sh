#!/bin/bash
# same-box ablation of the bf16 attention main loop (diagnostic builds with -DPETR_DIAG_BF16_NO_{EXP,STAGE,BARRIER})
for v in "" ${VARIANTS:-NO_EXP NO_STAGE NO_BARRIER}; do
  lib=$GRAFT_REPO_ROOT/petr_amd/lib/libpetr_hip${v:+_$v}.so
  echo "== ${v:-full}"
  PETR_HIP_LIB=$lib WLS="${WLS:-p4_1600}" SPLITS=${SPLITS:-4,8} bash $GRAFT_REPO_ROOT/scripts/attn_bf16_prof.sh | grep bf16_kernel
done
