cd $GRAFT_REPO_ROOT
python tools/attn_ragged_check.py 2>&1 | grep -v amdgpu.ids
echo "--- one block per workgroup"
AFHIP_ENC64_ONE_BLOCK_PER_WG=1 python tools/attn_ragged_check.py 2>&1 | grep -v amdgpu.ids
