#!/bin/bash
# Kernel timeline of one xhe_verify_batch call from host buffers (the last of two): tools/trace_e2e.sh <out.csv> [txs transfers]
out="$1"; n="${2:-100000}"; t="${3:-8}"
root="$(pwd)"
rm -rf "$root/gpurun_out/trace"; mkdir -p "$root/gpurun_out/trace"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace -d "$root/gpurun_out/trace" -o run -- python3 "$root/tools/exp_e2e.py" "$n" "$t" 2 > "$root/gpurun_out/trace/run.log" 2>&1 || { tail -5 "$root/gpurun_out/trace/run.log"; exit 1; }
grep E2E "$root/gpurun_out/trace/run.log"
db=$(ls -t "$root"/gpurun_out/trace/*.db "$root"/gpurun_out/trace/*/*.db 2>/dev/null | head -1)
python3 "$root/tools/rocpd_tail.py" "$db" 40 > "$root/$out"
