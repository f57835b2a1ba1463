#!/bin/bash
# One rocprofv3 --pmc pass over a short bench.py run; per-kernel counter averages as CSV.
# usage: tools/prof_pmc.sh <COUNTER> <out.csv> [bench.py arguments...]
counter="$1"; out="$2"; shift; shift
root="$(pwd)"
dir="$root/gpurun_out/pmc_$counter"
rm -rf "$dir"; mkdir -p "$dir"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --pmc "$counter" -d "$dir" -o run -- python3 "$root/bench.py" --lean --steps 3 --warmup 1 "$@" > "$dir/bench.log" 2>&1 || exit 1
db=$(ls -t "$dir"/*.db "$dir"/*/*.db 2>/dev/null | head -1)
python3 "$root/tools/rocpd_pmc.py" "$db" > "$root/$out"
