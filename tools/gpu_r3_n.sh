cd $GRAFT_REPO_ROOT
L=audio-intelligence_amd/csrc
export AFHIP_ATTN_ENC8=0
cp $L/libafhip.so /tmp/base.so
TAG="base " timeout -k 10 200 python tools/attn_probe.py 2>&1 | grep "prescaled :"
for v in NOEXP NOADD NOCVT; do
  cp $L/alt/libafhip_$v.so $L/libafhip.so
  TAG="$v" timeout -k 10 200 python tools/attn_probe.py 2>&1 | grep "prescaled :"
done
cp /tmp/base.so $L/libafhip.so
TAG="base " timeout -k 10 200 python tools/attn_probe.py 2>&1 | grep "prescaled :"
