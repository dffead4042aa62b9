R=$GRAFT_REPO_ROOT; cd $R; export PYTHONPATH=$R
python scripts/r6/probe_norm.py
