cd $GRAFT_REPO_ROOT
run() { name=$1; shift; env "$@" timeout -k 10 300 python -u bench.py --steps 20 --warmup 3 --lean --value-basis resident > gpurun_out/ab_$name.log 2>&1 || exit 1; python -c "
import json,sys
d=json.loads(open('gpurun_out/ab_$name.log').read().strip().splitlines()[-1]); print('$name', d['ms_per_step'], d['stages_ms'], 'secondary', d['secondary']['ms_per_step'])"; }
run occ2_512 XHE_DECODE_OCC=2
run occ4_512 XHE_DECODE_OCC=4
run occ4_1024 XHE_DECODE_OCC=4 XHE_DECODE_MAX_BLOCKS=1024
run occ4_2048 XHE_DECODE_OCC=4 XHE_DECODE_MAX_BLOCKS=2048
run occ2_1024 XHE_DECODE_OCC=2 XHE_DECODE_MAX_BLOCKS=1024
