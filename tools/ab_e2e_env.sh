#!/bin/bash
# End-to-end call time (tools/exp_e2e.py, headline batch) under several environments: tools/ab_e2e_env.sh "<VAR=val ...>" "<VAR=val ...>" ...
# ("-" = the plain environment).  One process per setting; each prints its calls.
for envs in "$@"; do
  echo "== $envs"
  if [ "$envs" = "-" ]; then timeout -k 10 200 python3 tools/exp_e2e.py 100000 8 5 2>&1 | grep E2E || exit 1
  else env $envs timeout -k 10 200 python3 tools/exp_e2e.py 100000 8 5 2>&1 | grep E2E || exit 1; fi
done
