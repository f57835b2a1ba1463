#!/bin/bash
# A/B of one environment switch on one box: tools/ab_env.sh NAME value1 value2 ...   (headline batch, two rounds)
cd "$GRAFT_REPO_ROOT"
name=$1; shift
for rep in 1 2; do
  for v in "$@"; do
    env_arg="$name=$v"
    env $env_arg timeout -k 10 300 python -u bench.py --lean --value-basis resident --steps 30 --warmup 5 > gpurun_out/ab_env.json 2> gpurun_out/ab_env.err || { tail -3 gpurun_out/ab_env.err; exit 1; }
    python - "$env_arg" "$rep" <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab_env.json"))
sd = d["stage_detail_ms"]
print(sys.argv[1], "rep", sys.argv[2], "ms_per_step", d["ms_per_step"], d["stages_ms"], "sig", sd.get("signatures"), "sort", [sd.get(k) for k in ("msm_histogram", "msm_scan", "msm_scatter", "msm_wait_points", "msm_merge", "msm_reduce")], flush=True)
PY
  done
done
