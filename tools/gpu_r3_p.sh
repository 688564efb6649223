cd $GRAFT_REPO_ROOT
L=audio-intelligence_amd/csrc
set -e
cp $L/libafhip.so /tmp/s2.so
echo "== sched 2 (current)"; timeout -k 10 200 python tools/gemm_bench.py 32 2>&1 | grep -v "square\|amdgpu.ids" | cut -c1-80
cp $L/alt/libafhip_s3.so $L/libafhip.so
timeout -k 10 300 python tools/gemm_pp_check.py check 2>&1 | tail -8
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_bf16.py -q -m gpu -k "gemm" 2>&1 | tail -2
echo "== sched 3 (two phases)"; timeout -k 10 200 python tools/gemm_bench.py 32 2>&1 | grep -v "square\|amdgpu.ids" | cut -c1-80
cp /tmp/s2.so $L/libafhip.so
echo "== sched 2 again"; timeout -k 10 200 python tools/gemm_bench.py 32 2>&1 | grep -v "square\|amdgpu.ids" | cut -c1-80
cp $L/alt/libafhip_s3.so $L/libafhip.so
echo "== sched 3 again"; timeout -k 10 200 python tools/gemm_bench.py 32 2>&1 | grep -v "square\|amdgpu.ids" | cut -c1-80
echo "== headline, sched 3"; timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu --no-decode --no-extra-legs --no-ceiling 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['roofline']['frac'])"
cp /tmp/s2.so $L/libafhip.so
echo "== headline, sched 2"; timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu --no-decode --no-extra-legs --no-ceiling 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['roofline']['frac'])"
