#!/bin/bash
# diagnostic: build petr_amd/lib/libpetr_hip_<tag>.so from the current tree (A/B timing on one box via PETR_HIP_LIB).
#   scripts/ab_build.sh tune -DPETR_TUNING_ENV      the PETR_* tuning switches read the environment (common.h petr_tune);
#                                                    the product library ignores them
#   scripts/ab_build.sh stamps -DPETR_DIAG_BWD_STAMPS
set -e
tag=$1; shift
make -C petr_amd/csrc OUT=../lib/libpetr_hip_$tag.so OBJDIR=../lib/obj_$tag EXTRA="$*" -j8 2>&1 | grep -E " error|Error" || true
ls -la petr_amd/lib/libpetr_hip_$tag.so
