set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_kernels.py tests/test_gpu_bf16.py -q -m gpu > gpurun_out/r3i_tests.log 2>&1; echo "tests rc=$?"
tail -5 gpurun_out/r3i_tests.log
