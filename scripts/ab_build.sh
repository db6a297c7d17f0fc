#!/bin/bash
# diagnostic: build petr_amd/lib/libpetr_hip_<tag>.so from the current tree (A/B timing on one box via PETR_HIP_LIB)
set -e
tag=$1
make -C petr_amd/csrc OUT=../lib/libpetr_hip_$tag.so OBJDIR=../lib/obj_$tag -j8 2>&1 | grep -E " error|Error" || true
ls -la petr_amd/lib/libpetr_hip_$tag.so
