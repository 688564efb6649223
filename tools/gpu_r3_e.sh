set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python tools/attn_var_probe.py > gpurun_out/r3e_var.log 2>&1; cat gpurun_out/r3e_var.log | tail -9
for v in 1 2 3; do AFHIP_ENC64_VAR=$v timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "prescaled" 2>&1 | tail -1 | sed "s/^/var $v tests: /"; done
