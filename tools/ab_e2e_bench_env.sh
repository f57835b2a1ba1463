#!/bin/bash
# The call from host buffers (bench's timed loop) under values of one switch: tools/ab_e2e_bench_env.sh NAME v1 v2 ...  ("-" = unset); prints the call's
# time and the accumulation's time per launch
cd "$GRAFT_REPO_ROOT"
name="$1"; shift
for v in "$@"; do
  if [ "$v" = "-" ]; then timeout -k 10 300 python -u bench.py --lean --steps 15 --warmup 4 > gpurun_out/ab_e2e_bench.json 2> gpurun_out/ab_e2e_bench.err || { tail -3 gpurun_out/ab_e2e_bench.err; exit 1; }
  else env $name=$v timeout -k 10 300 python -u bench.py --lean --steps 15 --warmup 4 > gpurun_out/ab_e2e_bench.json 2> gpurun_out/ab_e2e_bench.err || { tail -3 gpurun_out/ab_e2e_bench.err; exit 1; }; fi
  python - "$name" "$v" <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab_e2e_bench.json"))
print(sys.argv[1], sys.argv[2], "call ms", d["ms_per_step"], "accumulate ms per launch", d["roofline"]["kernel_ms"], flush=True)
PY
done
