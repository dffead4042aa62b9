R=$GRAFT_REPO_ROOT; cd $R; export PYTHONPATH=$R
echo "== prev"; IPOKE_LIB_PATH=$R/ipoke_amd/libipoke_prev.so python scripts/r6/probe_norm.py
for pos in 256 512 1024; do echo "== new pos $pos"; IPOKE_GN_APPLY_POS=$pos python scripts/r6/probe_norm.py; done
