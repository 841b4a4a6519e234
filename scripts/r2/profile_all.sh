#!/bin/bash
# round-2 profiles of HEAD for the three single-GPU workloads + bench lines (copied to profiles/round2/ by hand afterwards)
cd "$GRAFT_REPO_ROOT"
for wl in c2 c3 c5; do
  bash scripts/r2/profile.sh r2_$wl $wl > gpurun_out/prof_r2_$wl.log 2>&1 || { tail -5 gpurun_out/prof_r2_$wl.log; exit 1; }
  tail -3 gpurun_out/prof_r2_$wl.log
done
for wl in c2 c3 c5; do
  timeout -k 10 300 python bench.py --workload $wl --no-cpu-baseline 2>/dev/null | grep '"metric"' > gpurun_out/bench_r2_$wl.json || exit 1
done
timeout -k 10 300 python bench.py --workload c2 --precision bf16 --no-cpu-baseline 2>/dev/null | grep '"metric"' > gpurun_out/bench_r2_c2_bf16.json
# K sweep of the reference's timing script (launch_job/atari/launch_time.sh:13-27: K in 1, 4, 9, 49)
for K in 1 4 9 49; do
  timeout -k 10 300 python bench.py --workload c2 --K $K --no-cpu-baseline --steps 2000 --warmup 2000 2>/dev/null | grep '"metric"' > gpurun_out/bench_r2_c2_K$K.json || exit 1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/bench_r2_*.json")):
    d=json.loads(open(f).read()); print(f, round(d["value"],1), "steps/s", round(d["roofline"]["frac"],4))
PY
