#!/bin/bash
# Stage times of the resident verification for library builds on one box: tools/ab_stages.sh A B ...  (tools/variants/libxhe_<X>.so)
cd "$GRAFT_REPO_ROOT"
for v in "$@"; do
  XHE_LIBRARY="$GRAFT_REPO_ROOT/tools/variants/libxhe_$v.so" timeout -k 10 300 python -u bench.py --value-basis resident --no-secondary --no-pipelined --cpu-sample 0 --steps 30 --warmup 5 > gpurun_out/ab_stages.json 2> gpurun_out/ab_stages.err || { tail -3 gpurun_out/ab_stages.err; exit 1; }
  python - "$v" <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab_stages.json"))
r = d.get("roofline") or {}
print("variant", sys.argv[1], "ms_per_step", d["ms_per_step"], "accumulate_ms", r.get("kernel_ms"), "terms_acc", r.get("terms_accumulated"), "entries", r.get("entries_accumulated"))
print("   ", json.dumps(d.get("stage_detail_ms")), flush=True)
PY
done
