set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/attn_probe.py 2>&1 | grep "prescaled" | sed 's/^/persistent, no M0 restore: /'
AFHIP_ENC64_ONE_BLOCK_PER_WG=1 python tools/attn_probe.py 2>&1 | grep "prescaled :" | sed 's/^/one block per workgroup: /'
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "attention" 2>&1 | tail -1
AFHIP_ENC64_ONE_BLOCK_PER_WG=1 timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "attention" 2>&1 | tail -1
