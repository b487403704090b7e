#!/bin/bash
# the whole GPU gate, then the judged set of a round (tools/final_profiles.sh): what is run last, on the final tree (BUILD="round N final")
cd "$(dirname "${BASH_SOURCE[0]}")/.."
export GRAFT_REPO_ROOT=$PWD
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r6_final_pytest.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r6_final_pytest.txt
tail -6 gpurun_out/r6_final_pytest.txt
python3 tools/race_hunt.py 2500 7 2>&1 | grep -v amdgpu.ids > gpurun_out/r6_final_race_hunt.txt; tail -3 gpurun_out/r6_final_race_hunt.txt   # ~50 000 small calls against known answers (NOTES_r6 section 15)
BUILD="${BUILD:-round 6 final}" bash tools/final_profiles.sh 2>&1 | tail -12
