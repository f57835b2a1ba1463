#!/bin/bash
# Kernel timeline of balance-decoding calls: tools/trace_ecdlp.sh <out.csv> [table_bits=28]   (the last 8 ms of tools/exp_ecdlp_tables.py)
out="$1"; b="${2:-28}"; root="$(pwd)"
rm -rf "$root/gpurun_out/trace2"; mkdir -p "$root/gpurun_out/trace2"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d "$root/gpurun_out/trace2" -o run -- python3 "$root/tools/exp_ecdlp_tables.py" 40 "$b" > "$root/gpurun_out/trace2/run.log" 2>&1 || { tail -5 "$root/gpurun_out/trace2/run.log"; exit 1; }
cat "$root/gpurun_out/trace2/run.log" | tail -2
db=$(ls -t "$root"/gpurun_out/trace2/*.db "$root"/gpurun_out/trace2/*/*.db 2>/dev/null | head -1)
python3 "$root/tools/rocpd_tail.py" "$db" 8 > "$root/$out"
