cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_config5.py tests/test_gpu_kernels.py tests/test_gpu_bf16.py -q -m gpu -x -s -k "fp8 or e4m3" 2>&1 | grep -v amdgpu.ids | grep -E "token match|passed|failed|Error|assert" | tail -8
timeout -k 10 600 python tools/enc_fp8_static_probe.py 2>&1 | grep -v amdgpu.ids
timeout -k 10 300 python tools/mel_stamps.py 2>&1 | grep -v amdgpu.ids | tail -15
