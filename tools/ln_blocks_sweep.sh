export TMPDIR=/tmp; mkdir -p gpurun_out/r3r
for cap in 2048 1024 512 256; do
QT_LN_BLOCKS=$cap rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3r/ln$cap -o ln -- python3 tools/bench_ln.py > gpurun_out/r3r/ln$cap.log 2>&1
python - <<PY
import csv, collections
rows = list(csv.DictReader(open("gpurun_out/r3r/ln$cap/ln_kernel_trace.csv")))
v = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "qt_add_ln_bwd" in r["Kernel_Name"]]
# launches in order: 33 per config (3 warm + 30), four configs
n = len(v) // 4
print("cap $cap:", ["%.1f" % sorted(v[i*n:(i+1)*n])[n//2] for i in range(4)], "us (median per config: 4 addends, 2 addends, 15360x1, 256 rows)")
PY
rm -rf gpurun_out/r3r/ln$cap
done
