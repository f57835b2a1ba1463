#!/bin/bash
# Resident step time over the values of one environment switch for a workload: tools/ab_env_args.sh "<bench args>" NAME v1 v2 ...  ("-" = unset)
cd "$GRAFT_REPO_ROOT"
args="$1"; name="$2"; shift; shift
for v in "$@"; do
  if [ "$v" = "-" ]; then timeout -k 10 300 python -u bench.py --lean --value-basis resident --steps 20 --warmup 5 $args > gpurun_out/ab_env_args.json 2> gpurun_out/ab_env_args.err || { tail -3 gpurun_out/ab_env_args.err; exit 1; }
  else env $name=$v timeout -k 10 300 python -u bench.py --lean --value-basis resident --steps 20 --warmup 5 $args > gpurun_out/ab_env_args.json 2> gpurun_out/ab_env_args.err || { tail -3 gpurun_out/ab_env_args.err; exit 1; }; fi
  python - "$name" "$v" <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab_env_args.json"))
print(sys.argv[1], sys.argv[2], "ms_per_step", d["ms_per_step"], flush=True)
PY
done
