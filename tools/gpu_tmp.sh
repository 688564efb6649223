cd $GRAFT_REPO_ROOT
L=audio-intelligence_amd/csrc
set -e
cp $L/libafhip.so /tmp/new.so
for r in 1 2; do
echo "== new (dead half-tiles skip MFMAs)"; timeout -k 10 200 python tools/gemm_bench.py 32 2>&1 | grep -v "square\|amdgpu.ids" | cut -c1-80
cp $L/alt/libafhip_old.so $L/libafhip.so
echo "== old"; timeout -k 10 200 python tools/gemm_bench.py 32 2>&1 | grep -v "square\|amdgpu.ids" | cut -c1-80
cp /tmp/new.so $L/libafhip.so
done
timeout -k 10 300 python tools/gemm_pp_check.py check 2>&1 | tail -8
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_bf16.py tests/test_gpu_encoder.py -q -m gpu -k "gemm or ragged or encoder" 2>&1 | tail -2
