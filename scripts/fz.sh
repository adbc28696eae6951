for envs in "" "FWGPU_CTL_AHEAD=0" "FWGPU_RT_PERSIST=0" "FWGPU_CTL_AHEAD=0 FWGPU_RT_PERSIST=0"; do
for seed in 91 196 384; do
r=$(env $envs FWGPU_FUZZ_SEEDS=500 timeout 300 python -m pytest "tests/test_fuzz_gpu.py::test_random_graph_and_messages_every_plan_bit_exact[$seed]" -m gpu -q -x 2>&1 | grep -E "^E.*differ|passed|failed" | head -2 | tr '\n' ' ')
echo "[$envs] seed $seed: $r"
done; done
