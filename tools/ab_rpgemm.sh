#!/bin/bash
# A/B on one box: range-proof column sums through the matrix cores against the element-by-element kernel (headline batch)
set -e
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
  for mode in 1099511627776 512; do
    XHE_RP_GEMM_MIN_ROWS=$mode timeout -k 10 300 python -u bench.py --lean --value-basis resident --steps 30 --warmup 5 > gpurun_out/ab_rpgemm_${mode}_${rep}.json 2> gpurun_out/ab_rpgemm.err
    python - "$mode" "$rep" <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab_rpgemm_%s_%s.json" % (sys.argv[1], sys.argv[2])))
print("min_rows", sys.argv[1], "rep", sys.argv[2], "ms_per_step", d["ms_per_step"], d["stages_ms"], flush=True)
PY
  done
done
