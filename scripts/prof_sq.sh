# usage: bash scripts/prof_sq.sh <tag> [bench args...]  — SQ instruction-mix / stall counters of the fwgpu kernels,
# two --pmc passes of <= 8 SQ counters each (run on the GPU box via gpurun)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
pass=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU" \
            "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM"; do
  pass=$((pass+1))
  out=$GRAFT_REPO_ROOT/gpurun_out/sq_${tag}_$pass
  timeout 120 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out -o $tag -- python $GRAFT_REPO_ROOT/bench.py --lean --no-kernel-timing --steps 4 --warmup 1 "$@" > $out.log 2>&1
  f=$(find $out -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"].split("(")[0]
    if "fwgpu" in k:
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-24s launches=%d avg=%.4g" % (c, len(v), sum(v) / len(v)))
PY
done
