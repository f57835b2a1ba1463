#!/bin/bash
# Kernel timeline of single-transaction verifications: tools/trace_single.sh <out.csv>   (the last millisecond and a half of the run: resident calls)
out="$1"; root="$(pwd)"
rm -rf "$root/gpurun_out/trace1"; mkdir -p "$root/gpurun_out/trace1"
cd /tmp && export TMPDIR=/tmp
XHE_ENQUEUE_TIMES=1 timeout -k 10 300 rocprofv3 --kernel-trace -d "$root/gpurun_out/trace1" -o run -- python3 "$root/tools/exp_single.py" 1 20 > "$root/gpurun_out/trace1/run.log" 2>&1 || { tail -5 "$root/gpurun_out/trace1/run.log"; exit 1; }
grep "single transaction" "$root/gpurun_out/trace1/run.log"
grep "xhe enqueue" "$root/gpurun_out/trace1/run.log" | tail -8
db=$(ls -t "$root"/gpurun_out/trace1/*.db "$root"/gpurun_out/trace1/*/*.db 2>/dev/null | head -1)
python3 "$root/tools/rocpd_tail.py" "$db" 1.6 > "$root/$out"
