# usage: bash scripts/prof_pmc.sh <tag>   — HBM traffic counters, one --pmc pass each (guide: MI355X_MICROARCH.md §HBM)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  out=$GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_$ctr
  timeout 120 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out -o $tag -- python $GRAFT_REPO_ROOT/bench.py --lean --no-kernel-timing --steps 6 --warmup 2 "$@" > $out.log 2>&1
  f=$(find $out -name "*counter_collection.csv" | head -1)
  python - "$f" $ctr <<'PY'
import csv, sys, collections
f, ctr = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for row in csv.DictReader(open(f)):
    if row.get("Counter_Name") == ctr:
        acc[row["Kernel_Name"].split("(")[0]].append(float(row["Counter_Value"]))
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    if "fwgpu" in k:
        print("%s %s launches=%d avg_per_launch=%.1f" % (ctr, k, len(v), sum(v) / len(v)))
PY
done
