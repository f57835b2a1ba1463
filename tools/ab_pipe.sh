#!/bin/bash
# A/B of library builds on the pipelined path: tools/ab_pipe.sh A B ...  (tools/variants/libxhe_<X>.so in turn, two rounds)
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
  for v in "$@"; do
    XHE_LIBRARY="$GRAFT_REPO_ROOT/tools/variants/libxhe_$v.so" timeout -k 10 300 python -u bench.py --cpu-sample 0 --no-secondary --resident-steps 0 --steps 10 --warmup 3 > gpurun_out/ab_pipe.json 2> gpurun_out/ab_pipe.err || { tail -3 gpurun_out/ab_pipe.err; exit 1; }
    python - "$v" "$rep" <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab_pipe.json"))
e = d["pipelined"]
print("variant", sys.argv[1], "rep", sys.argv[2], "single", d["ms_per_step"], "pipelined", e["ms_per_batch"], "steady", e["steady_ms_per_batch"], flush=True)
PY
  done
done
