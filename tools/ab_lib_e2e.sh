#!/bin/bash
# A/B of library builds on the call from host buffers: tools/ab_lib_e2e.sh A B ...  (tools/variants/libxhe_<X>.so in turn, two rounds; the bench's timed loop
# with its accumulation statistics)
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
  for v in "$@"; do
    XHE_LIBRARY="$GRAFT_REPO_ROOT/tools/variants/libxhe_$v.so" timeout -k 10 300 python -u bench.py --lean --steps 20 --warmup 5 > gpurun_out/ab_lib_e2e.json 2> gpurun_out/ab_lib_e2e.err || { tail -3 gpurun_out/ab_lib_e2e.err; exit 1; }
    python - "$v" "$rep" <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab_lib_e2e.json"))
print("variant", sys.argv[1], "rep", sys.argv[2], "call ms", d["ms_per_step"], "accumulate ms per launch", d["roofline"]["kernel_ms"], flush=True)
PY
  done
done
