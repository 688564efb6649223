set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "attention" > gpurun_out/r3b_attn_tests.log 2>&1; echo "attn tests rc=$?"
tail -3 gpurun_out/r3b_attn_tests.log
python tools/attn_probe.py > gpurun_out/r3b_attn_probe.log 2>&1; cat gpurun_out/r3b_attn_probe.log
for m in 7 6 14 5 3 15; do
  AFHIP_FP8_MASK=$m python -m pytest tests/test_gpu_config5.py -q -m gpu -s -k "fp8_encoder" 2>&1 | grep -E "fp8 encoder vs|7B-width sample|passed|failed" | sed "s/^/mask=$m: /"
done > gpurun_out/r3b_fp8_masks.log 2>&1
cat gpurun_out/r3b_fp8_masks.log
rocprofv3 -L > gpurun_out/r3b_counters.txt 2>&1; grep -c "" gpurun_out/r3b_counters.txt
