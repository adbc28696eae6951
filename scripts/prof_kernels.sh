# usage: bash scripts/prof_kernels.sh <tag> [bench args...]   (run on the GPU box via gpurun)
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o $tag -- python $GRAFT_REPO_ROOT/bench.py --lean "$@" > $out.log 2>&1
find $out -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -12 {}'
