cd /root/repo; export TMPDIR=/tmp
rm -rf gpurun_out/vth; HOSTBUF=$1 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/vth -o t -- python3 benchmarks/verify_timing.py 4096 > gpurun_out/vth.log 2>&1
f=$(find gpurun_out/vth -name "*kernel_trace.csv" | head -n 1); m=$(find gpurun_out/vth -name "*memory_copy_trace.csv" | head -n 1)
python3 - "$f" "$m" <<'PY'
import csv,sys
rows=[]
for r in csv.DictReader(open(sys.argv[1])): rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"].split("(")[0].replace("bppp::","")[:30]))
try:
    for r in csv.DictReader(open(sys.argv[2])): rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),"COPY "+r.get("Direction","")+" "+r.get("Bytes","")))
except Exception as e: print("no copy trace",e)
rows.sort()
idx=[i for i,r in enumerate(rows) if "k_rp_rho" in r[2]]
last=idx[-1]; prev=idx[-2]
# start of last call = first row after previous call's last kernel (copyBuffer after reduce_tail)
start=prev
while start<last and "decode_points" not in rows[start][2] : start+=1
start=max(0,start-6)
t0=rows[start][0]
for r in rows[start:last+2]:
    print(f"{(r[0]-t0)/1e6:8.3f} {(r[1]-r[0])/1e6:7.3f} {r[2]}")
PY
