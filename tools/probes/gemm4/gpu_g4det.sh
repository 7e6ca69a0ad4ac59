#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; rm -f gpurun_out/r2_g4det.txt
run() { echo "== $*" | tee -a gpurun_out/r2_g4det.txt; env "$@" timeout -k 10 200 python3 tools/g4_repeat.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r2_g4det.txt; }
run PTTS_G4_DBG=0
run PTTS_G4_DBG=1
run PTTS_G4_DBG=4
run PTTS_G4_DBG=2
run PTTS_G4_PAD=40000
