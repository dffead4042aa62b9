R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
export TMPDIR=/tmp
for m in base reuse; do
  python scripts/r6/probe_boundary.py $m 2>/dev/null | tail -1
  IPOKE_PROBE_NO_READY_JOIN=1 python scripts/r6/probe_boundary.py $m 2>/dev/null | tail -1
  IPOKE_PROBE_SKIP_ADAM=1 python scripts/r6/probe_boundary.py $m 2>/dev/null | tail -1
done
