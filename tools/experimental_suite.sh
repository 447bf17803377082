#!/bin/bash
# The opt-in paths built at the end of round 4, in one GPU call (~17 min):
#  1. the whole GPU suite with the row-resident decoder chains switched on for every engine (FAR3D_FUSED_ROWS=1) and the default
#     bench line with them -- the evidence needed before engine.fused_rows becomes the default;
#  2. the gated tests of the A/B variants (8-part attention, camera-group latency runner) and the latency bench line.
#   /usr/local/graft/bin/gpurun --timeout 1300 -- 'bash tools/experimental_suite.sh r5a'
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
O=gpurun_out/ev_${1:-r5a}
mkdir -p "$O"
FAR3D_FUSED_ROWS=1 timeout 900 python -m pytest tests -q -m gpu --durations=10 > "$O/pytest_gpu_fused_rows.txt" 2>&1; echo "rc=$?" >> "$O/pytest_gpu_fused_rows.txt"
tail -4 "$O/pytest_gpu_fused_rows.txt"
timeout 170 python bench.py --fused-rows > "$O/bench_fused_rows.json" 2> "$O/bench_fused_rows.err"; echo "bench rc=$?"
tail -c 600 "$O/bench_fused_rows.json"
# experimental A/B variants (not part of the default suite): the 8-part bf16 attention kernel, with its device time
FAR3D_TEST_EXPERIMENTAL=1 timeout 120 python -m pytest tests/test_attn_norm_gpu.py -q -s -k eight_key_parts > "$O/pytest_attn_parts8.txt" 2>&1; tail -4 "$O/pytest_attn_parts8.txt"
FAR3D_TEST_EXPERIMENTAL=1 timeout 200 python -m pytest tests/test_latency_gpu.py -q -s > "$O/pytest_latency_groups.txt" 2>&1; tail -4 "$O/pytest_latency_groups.txt"
timeout 170 python bench.py --latency-groups 2 --fused-rows --no-cpu-baseline --no-in-tolerance --steps 40 > "$O/bench_latency_groups.json" 2> "$O/bench_latency_groups.err"; echo "latency bench rc=$?"
