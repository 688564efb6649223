set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "attention" > gpurun_out/r3d_attn_tests.log 2>&1; echo "attn tests rc=$?"
tail -15 gpurun_out/r3d_attn_tests.log
timeout -k 10 120 python tools/attn_probe.py > gpurun_out/r3d_attn_probe.log 2>&1; cat gpurun_out/r3d_attn_probe.log | tail -6
AFHIP_ATTN_ENC64=0 timeout -k 10 120 python tools/attn_probe.py 2>&1 | tail -5 | sed 's/^/enc64 off: /'
