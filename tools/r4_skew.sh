#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python -m pytest tests/test_gpu_1_parity.py tests/test_gpu_2_stages.py -x -q -m gpu 2>&1 | tail -4
python tools/adversarial_timing.py 2>&1 | grep -v amdgpu.ids
python tools/ab_libs.py --rounds 2 base
python tools/ab_libs.py --rounds 2 base -- --log-n 17
