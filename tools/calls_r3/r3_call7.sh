# round-3 GPU call 7: aggregation variant 12 (scatter-add tap merge), conv tiles for the per-rank shapes of camera-sharded runs
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c7; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_sampling_gpu.py -q 2>&1 | tail -6 | tee $O/pytest_sampling.txt
timeout 300 python tools/bench_kernels.py --iters 50 --out $O/kernels.jsonl > /dev/null 2>&1; grep -v '"v3' $O/kernels.jsonl | cut -c1-120
timeout 300 python tools/agg_phase_times.py 1544 12 > $O/agg_phase_v12.txt 2>&1; tail -13 $O/agg_phase_v12.txt
for n in 1 2 4; do NCAM=$n timeout 600 python tools/tune_conv.py > $O/tune_bf16_n$n.log 2>&1; tail -3 $O/tune_bf16_n$n.log | cut -c1-200; done
python - <<'PY'
import json, os
base = json.load(open("far3d_amd/data/tuning_mi355x.json"))
for n in (1, 2, 4):
    p = "gpurun_out/tuning_mi355x_n%d.json" % n
    if os.path.exists(p):
        base.update(json.load(open(p)))
json.dump(base, open("far3d_amd/data/tuning_mi355x.json", "w"), indent=0, sort_keys=True)
json.dump(base, open("gpurun_out/tuning_mi355x_merged.json", "w"), indent=0, sort_keys=True)
print("merged entries:", len(base))
PY
timeout 300 python tools/stage_times.py bf16 2>&1 | tail -16 | tee $O/stage_times_bf16_tuned.txt
