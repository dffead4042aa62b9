mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/encprof -o enc -- python $R/scripts/probe_enc.py > $R/gpurun_out/enc_probe.txt 2>&1
cd $R
f=$(find /tmp/encprof -name "*kernel_stats.csv" | head -1)
cp $f gpurun_out/enc_kernel_stats.csv
t=$(find /tmp/encprof -name "*kernel_trace.csv" | head -1)
python - <<PY
import csv
rows=list(csv.DictReader(open("$t")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last make_flow_input call: take the final N dispatches where N = len/13
n=len(rows)//13
last=rows[-n:]
t0=int(last[0]["Start_Timestamp"])
with open("gpurun_out/enc_last_call.txt","w") as f:
    for r in last:
        f.write(f'{(int(r["Start_Timestamp"])-t0)/1e3:9.1f} {(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3:8.1f}  grid={r.get("Grid_Size_X","?")},{r.get("Grid_Size_Y","?")},{r.get("Grid_Size_Z","?")} wg={r.get("Workgroup_Size_X","?")} {r["Kernel_Name"][:110]}\n')
PY
head -30 gpurun_out/enc_kernel_stats.csv | cut -c1-200
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
