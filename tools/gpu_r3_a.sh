set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_config5.py tests/test_gpu_generation.py tests/test_gpu_llm.py -x -q -m gpu -s > gpurun_out/r3a_tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r3a_tests.log
tail -5 gpurun_out/r3a_tests.log
AFHIP_BENCH_REHEARSAL=1 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu --no-decode --no-extra-legs --no-ceiling > gpurun_out/r3a_rehearsal.log 2>&1; echo "rehearsal rc=$?"
tail -c 1500 gpurun_out/r3a_rehearsal.log
bash tools/prof_r03.sh r03a > gpurun_out/r3a_prof.log 2>&1; echo "prof rc=$?"
tail -40 gpurun_out/r3a_prof.log
