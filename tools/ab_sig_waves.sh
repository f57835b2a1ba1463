#!/bin/bash
# Resident step time over XHE_SIG_MAX_WAVES for a workload: tools/ab_sig_waves.sh "<bench args>" v1 v2 ...
cd "$GRAFT_REPO_ROOT"
args="$1"; shift
for v in "$@"; do
  XHE_SIG_MAX_WAVES=$v timeout -k 10 300 python -u bench.py --lean --value-basis resident --steps 20 --warmup 5 $args > gpurun_out/ab_sig.json 2> gpurun_out/ab_sig.err || { tail -3 gpurun_out/ab_sig.err; exit 1; }
  python - "$v" <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab_sig.json"))
print("XHE_SIG_MAX_WAVES", sys.argv[1], "ms_per_step", d["ms_per_step"], flush=True)
PY
done
