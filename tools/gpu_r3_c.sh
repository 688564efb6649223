set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_encoder.py -x -q -m gpu -k "log_mel or preprocess or mel" > gpurun_out/r3c_mel_tests.log 2>&1; echo "mel tests rc=$?"; tail -3 gpurun_out/r3c_mel_tests.log
python tools/mel_one.py > gpurun_out/r3c_mel_one.log 2>&1; tail -5 gpurun_out/r3c_mel_one.log
python tools/attn_stamps.py > gpurun_out/r3c_attn_stamps.log 2>&1; cat gpurun_out/r3c_attn_stamps.log | tail -24
bash tools/pmc_r03.sh r03a > gpurun_out/r3c_pmc.log 2>&1; tail -40 gpurun_out/r3c_pmc.log
