cd $GRAFT_REPO_ROOT
L=audio-intelligence_amd/csrc
cp $L/libafhip.so /tmp/base.so
TAG="base     " timeout -k 10 200 python tools/attn_probe.py 2>&1 | grep "prescaled :"
cp $L/alt/libafhip_HALFSYNC.so $L/libafhip.so
TAG="half sync" timeout -k 10 200 python tools/attn_probe.py 2>&1 | grep "prescaled :"
cp /tmp/base.so $L/libafhip.so
TAG="base     " timeout -k 10 200 python tools/attn_probe.py 2>&1 | grep "prescaled :"
