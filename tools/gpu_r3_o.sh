cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_config5.py -q -m gpu -x -s -k "fp8" 2>&1 | grep -v amdgpu.ids | tail -15
