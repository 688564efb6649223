set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/full_tests.log 2>&1; echo "tests rc=$?"
tail -6 gpurun_out/full_tests.log
timeout -k 10 400 python bench.py --steps 10 --warmup 3 > gpurun_out/full_bench.json 2> gpurun_out/full_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/full_bench.json') if l.startswith('{')][-1])
print('value', d['value'], 'ms/step', d['ms_per_step'], 'frac', d['roofline']['frac'], 'avg_launch_ms', d['roofline']['avg_launch_ms'])
print('mel', d['stages']['mel_ms'], d['stages']['mel_roofline']['frac'], 'enc_ms', d['stages']['encoder_ms'])
print('fp8', d['stages']['encoder_fp8'])
print('mixed', d['stages']['encoder_mixed_lengths'])
print('decode', d['decode']['ms_per_step'], d['decode']['fp8_weights']['ms_per_step'], 'b16', d['decode_b16']['ms_per_step'], d['decode_b16']['fp8_weights']['ms_per_step'])
print('cpu', d['cpu_baseline'])
PY
