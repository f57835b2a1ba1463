#!/bin/bash
# FETCH_SIZE against a known byte count for the bucket accumulation's access shape: a timing run of tools/ubench_gather, then one
# rocprofv3 --pmc FETCH_SIZE pass over the same program; writes gpurun_out/<TAG>_gather_calibration.json.
# usage (on the GPU box, from the repository root): TAG=r03 bash tools/calibrate_gather.sh [log2_records] [entries_per_lane]
set -e
root="$(pwd)"
bits=${1:-24}; per_lane=${2:-32}
tag=${TAG:-run}
timeout -k 10 120 "$root/tools/ubench_gather" $bits $per_lane 5 > "$root/gpurun_out/${tag}_gather_timing.jsonl"
cat "$root/gpurun_out/${tag}_gather_timing.jsonl"
dir="$root/gpurun_out/pmc_gather"
rm -rf "$dir"; mkdir -p "$dir"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d "$dir" -o run -- "$root/tools/ubench_gather" $bits $per_lane 2 > "$dir/run.log" 2>&1
db=$(ls -t "$dir"/*.db "$dir"/*/*.db 2>/dev/null | head -1)
python3 "$root/tools/rocpd_pmc.py" "$db" > "$root/gpurun_out/${tag}_pmc_fetch_size_gather.csv"
cat "$root/gpurun_out/${tag}_pmc_fetch_size_gather.csv"
python3 - "$root/gpurun_out/${tag}_gather_timing.jsonl" "$root/gpurun_out/${tag}_pmc_fetch_size_gather.csv" $bits $per_lane > "$root/gpurun_out/${tag}_gather_calibration.json" <<'PY'
import csv, json, sys
timing = {}
for line in open(sys.argv[1]):
    r = json.loads(line)
    timing[r["kernel"]] = r
reported = {}
for row in csv.DictReader(open(sys.argv[2])):
    if row["counter"] == "FETCH_SIZE":
        reported[row["kernel"]] = float(row["avg"]) * 1024.0
out = {"what": "FETCH_SIZE (rocprofv3, KiB per dispatch x 1024) against the bytes a launch of tools/ubench_gather must move: every record named once, "
               "pseudo-random order, table beyond the on-die cache", "log2_records": int(sys.argv[3]), "entries_per_lane": int(sys.argv[4]), "kernels": {}}
for name, t in timing.items():
    key = next((k for k in reported if k.replace("void ", "").replace(" ", "").startswith(name.split("<")[0]) and name.replace(" ", "") in k.replace(" ", "")), None)
    if key is None:
        key = next((k for k in reported if name.split("<")[0] in k and "<" not in name), None)
    rep = reported.get(key)
    out["kernels"][name] = {"entries": t["entries"], "useful_bytes": t["useful_bytes"], "line_bytes": t["line_bytes"], "ms": t["ms"], "useful_GBps": t["useful_GBps"],
                            "line_GBps": t["line_GBps"], "fetch_size_reported": rep,
                            "lines_per_reported": (t["line_bytes"] / rep) if rep else None, "useful_per_reported": (t["useful_bytes"] / rep) if rep else None}
g = out["kernels"].get("gather<6,128>", {})
out["fetch_size_factor"] = g.get("lines_per_reported")
out["fetch_size_factor_note"] = "bytes of the 128-byte lines a launch touches per byte FETCH_SIZE reports, for six 16-byte LDS-DMA loads per lane from a random 128-byte record: multiply the reported figure by it"
out["gather_line_GBps"] = g.get("line_GBps")
print(json.dumps(out, indent=1))
PY
cat "$root/gpurun_out/${tag}_gather_calibration.json"
