#!/bin/bash
# Kernel-time profile of one bench.py run on the GPU box: rocprofv3 --kernel-trace --stats, summary as CSV.
# usage: tools/prof_kernels.sh <out.csv> [bench.py arguments...]
out="$1"; shift
root="$(pwd)"
rm -rf "$root/gpurun_out/prof"; mkdir -p "$root/gpurun_out/prof"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d "$root/gpurun_out/prof" -o run -- python3 "$root/bench.py" --lean "$@" > "$root/gpurun_out/prof/bench.log" 2>&1 || exit 1
db=$(ls -t "$root"/gpurun_out/prof/*.db "$root"/gpurun_out/prof/*/*.db 2>/dev/null | head -1)
python3 "$root/tools/rocpd_stats.py" "$db" > "$root/$out"
python3 "$root/tools/rocpd_timeline.py" "$db" > "$root/${out%.csv}_timeline.csv"
