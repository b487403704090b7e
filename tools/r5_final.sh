#!/bin/bash
# round 5: the stage tests touched last, then the judged set (tools/final_profiles.sh)
cd "$(dirname "${BASH_SOURCE[0]}")/.."
export GRAFT_REPO_ROOT=$PWD
timeout 600 python -m pytest tests/test_gpu_2_stages.py -m gpu -x -q 2>&1 | tail -3
BUILD="round 5 final" bash tools/final_profiles.sh 2>&1 | tail -30
