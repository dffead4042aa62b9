mkdir -p gpurun_out
export TMPDIR=/tmp; R=$PWD; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c5 -- python $R/bench.py --config c5 --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
cp $(find /tmp/p_c5 -name "*kernel_stats.csv" | head -1) $R/gpurun_out/c5_now_kernel_stats.csv
t=$(find /tmp/p_c5 -name "*kernel_trace.csv" | head -1)
python - <<PY
import csv
rows=list(csv.DictReader(open("$t")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
n=len(rows)
out=[]
for r in rows[int(n*0.75):]:
    d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    if d>120: out.append(f'{d:9.1f} grid={r["Grid_Size_X"]},{r["Grid_Size_Y"]} wg={r["Workgroup_Size_X"]} {r["Kernel_Name"][:100]}')
open("$R/gpurun_out/c5_long_kernels.txt","w").write("\n".join(out))
PY
