set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "attention" > gpurun_out/r3f_attn_tests.log 2>&1; echo "attn tests rc=$?"
tail -5 gpurun_out/r3f_attn_tests.log
timeout -k 10 120 python tools/attn_probe.py > gpurun_out/r3f_attn_probe.log 2>&1; tail -4 gpurun_out/r3f_attn_probe.log
