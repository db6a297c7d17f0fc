#!/bin/bash
# stamps of the fp32 attention backward (diagnostic library built in the build container: see scripts/bwd32_stamps.py)
cd $GRAFT_REPO_ROOT
PETR_HIP_LIB=$GRAFT_REPO_ROOT/petr_amd/lib/libpetr_hip_stamps.so python scripts/bwd32_stamps.py 0.0 > gpurun_out/r3_stamps.txt 2>&1 &&
PETR_HIP_LIB=$GRAFT_REPO_ROOT/petr_amd/lib/libpetr_hip_stamps.so python scripts/bwd32_stamps.py 0.1 >> gpurun_out/r3_stamps.txt 2>&1
cat gpurun_out/r3_stamps.txt
