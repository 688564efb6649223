cd $GRAFT_REPO_ROOT
L=audio-intelligence_amd/csrc
set -e
cp $L/libafhip.so /tmp/s0.so
echo "== sched 0"; timeout -k 10 200 python tools/gemm_bench.py 32 2>&1 | grep -v "square\|amdgpu.ids" | cut -c1-80
cp $L/alt/libafhip_s2.so $L/libafhip.so
timeout -k 10 300 python tools/gemm_pp_check.py check 2>&1 | tail -8
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_bf16.py -q -m gpu -k "gemm" 2>&1 | tail -2
echo "== sched 2"; timeout -k 10 200 python tools/gemm_bench.py 32 2>&1 | grep -v "square\|amdgpu.ids" | cut -c1-80
cp /tmp/s0.so $L/libafhip.so
echo "== sched 0 again"; timeout -k 10 200 python tools/gemm_bench.py 32 2>&1 | grep -v "square\|amdgpu.ids" | cut -c1-80
cp $L/alt/libafhip_s2.so $L/libafhip.so
echo "== sched 2 again"; timeout -k 10 200 python tools/gemm_bench.py 32 2>&1 | grep -v "square\|amdgpu.ids" | cut -c1-80
