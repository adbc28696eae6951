# round 6: per-launch durations of the level executor on config 2's graph (rocprofv3 --kernel-trace), in launch order, last step
cd /tmp && export TMPDIR=/tmp
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/raw
out=$GRAFT_REPO_ROOT/gpurun_out/raw/r06_levels_trace
rm -rf $out; timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out -o t -- python $GRAFT_REPO_ROOT/bench.py --workload cfg2 --lean --contexts 1 --force-generic --steps 3 --warmup 2 > $out.log 2>&1
f=$(find $out -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "fwgpu" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for r in rows[-14:]:
    print("%-40s %8.1f us  grid %s" % (r["Kernel_Name"].split("(")[0].replace("void fwgpu::", ""), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size_X", "?") + "x" + r.get("Grid_Size_Y", "?")))
PY
