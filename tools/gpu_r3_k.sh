cd $GRAFT_REPO_ROOT
PROBE_BF16_ONLY= python tools/decode_probe.py 8 64 790 2>&1 | grep "B=8" | sed 's/^/combine launch: /'
AFHIP_DECODE_MERGE=1 python tools/decode_probe.py 8 64 790 2>&1 | grep "B=8" | sed 's/^/in-launch sc1 merge: /'
AFHIP_DECODE_MERGE=1 timeout -k 10 600 python -m pytest tests/test_gpu_llm.py tests/test_gpu_bf16.py tests/test_gpu_kernels.py -q -m gpu -k "decode or greedy or 7b or split" 2>&1 | tail -2
