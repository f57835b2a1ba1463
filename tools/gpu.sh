#!/bin/bash
# Runs one gpurun call, waiting and retrying only while the pool reports "no box free" (exit code 3:
# nothing ran, nothing was charged).  Any other outcome is returned as is.
# usage: tools/gpu.sh <logfile> <timeout-seconds> '<command>'
log="$1"; to="$2"; cmd="$3"
for attempt in $(seq 1 20); do
  /usr/local/graft/bin/gpurun --timeout "$to" -- "$cmd" > "$log" 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 45
done
exit 3
