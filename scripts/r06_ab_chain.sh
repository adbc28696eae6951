# round 6: same-box A/B of the chain plan's new shapes (FWGPU_BENCH_CHAIN_SHAPE: v gain, B / b biquads, D delay, p pan) and of builds of the
# two-biquad instantiation (libfwgpu_n3.so: make OBJDIR=_obj_n3 OUT=libfwgpu_n3.so EXTRA=-DCH_S2B_WAVE=6: the second recurrence on a wave of its own)
run() { lib=$1; shape=$2; shift 2; FWGPU_LIB=$PWD/firewheel_amd/csrc/$lib FWGPU_BENCH_CHAIN_SHAPE=$shape python bench.py --workload cfg3 --chain-reordered --lean --steps 30 "$@" 2>/dev/null | python scripts/benchline.py "$lib $shape $* skip=$FWGPU_CHAIN_SKIP"; }
timeout 900 python -m pytest tests/test_chain_grammar.py -m gpu -q 2>&1 | tail -3
python bench.py --workload cfg3 --lean --steps 30 2>/dev/null | python scripts/benchline.py "cfg3 plain"
for i in 1 2; do for l in libfwgpu.so; do [ -f firewheel_amd/csrc/$l ] && run $l vBbDp; done; done
run libfwgpu.so vBbDp --source-format i16
run libfwgpu.so vBDp
run libfwgpu.so DBbp
FWGPU_CHAIN_SKIP=32 run libfwgpu.so vBbDp
