#!/bin/bash
# A/B of library builds on one box: tools/ab_lib.sh A B[:ENV=value] ...  runs the headline batch with tools/variants/libxhe_<X>.so in turn (selected through XHE_LIBRARY; the in-tree library is not touched)
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
  for spec in "$@"; do
    v=${spec%%:*}; e=""; [ "$spec" != "$v" ] && e=${spec#*:}
    env XHE_LIBRARY="$GRAFT_REPO_ROOT/tools/variants/libxhe_$v.so" $e timeout -k 10 300 python -u bench.py --lean --value-basis resident --steps 30 --warmup 5 > gpurun_out/ab_lib.json 2> gpurun_out/ab_lib.err || { tail -3 gpurun_out/ab_lib.err; exit 1; }
    python - "$spec" "$rep" <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab_lib.json"))
print("variant", sys.argv[1], "rep", sys.argv[2], "ms_per_step", d["ms_per_step"], d["stages_ms"], flush=True)
PY
  done
done
