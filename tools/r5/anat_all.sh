cd "${GRAFT_REPO_ROOT:-/root/repo}"
for t in tm0 tmc tm8 tm4 tm1 tm3 tm12 tm15 tm16; do echo "== $t"; python tools/r5/attn_anatomy.py visrag_amd/libvisrag_hip_$t.so 2>&1 | grep -v amdgpu.ids | head -4 | tail -3; done
