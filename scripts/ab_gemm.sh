#!/bin/bash
# diagnostic: scripts/gemm_sweep.py under several library builds on ONE box
for tag in "$@"; do echo "== $tag"; PETR_HIP_LIB=$PWD/petr_amd/lib/libpetr_hip_$tag.so timeout -k 10 200 python3 scripts/gemm_sweep.py 2>/dev/null | grep -E "M= *(4224|25344|16384|5400)" ; done
