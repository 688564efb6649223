cd $GRAFT_REPO_ROOT
L=audio-intelligence_amd/csrc
set -e
cp $L/libafhip.so /tmp/v0.so
for v in 0 1 2 0 2; do
  if [ $v = 0 ]; then cp /tmp/v0.so $L/libafhip.so; else cp $L/alt/libafhip_sp$v.so $L/libafhip.so; fi
  TAG="spread $v" timeout -k 10 200 python tools/attn_probe.py 2>&1 | grep prescaled
done
cp $L/alt/libafhip_sp2.so $L/libafhip.so
timeout -k 10 900 python -m pytest tests/ -q -m gpu -k "attention or attn or ragged" 2>&1 | tail -2
cp $L/alt/libafhip_sp1.so $L/libafhip.so
timeout -k 10 900 python -m pytest tests/ -q -m gpu -k "attention or attn or ragged" 2>&1 | tail -2
