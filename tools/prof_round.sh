# One round of measurements on the GPU box: the bench line, then kernel statistics and the three counter passes (each a run of its
# own) for the timed step as the bench runs it (a call from host buffers: a launch per part) and for the resident verification (one
# launch over the whole batch), then the resident passes of the mixed and the 125 000 x 1 batches.  TAG names the files.
set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -u bench.py > gpurun_out/${TAG:-run}_bench_n1.json 2> gpurun_out/${TAG:-run}_bench_n1.err
tail -c 600 gpurun_out/${TAG:-run}_bench_n1.json
passes() {   # <file tag> <bench arguments...>
  ftag=$1; shift
  bash tools/prof_kernels.sh gpurun_out/${TAG:-run}_kernel_stats_${ftag}.csv --steps 3 --warmup 1 "$@"
  for c in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU; do
    n=$(echo $c | tr 'A-Z' 'a-z'); [ $c = SQ_INSTS_VALU ] && n=valu
    bash tools/prof_pmc.sh $c gpurun_out/${TAG:-run}_pmc_${n}_${ftag}.csv "$@"
    echo "pmc $c $ftag done"
  done
}
passes 100000tx_t8
passes 100000tx_t8_resident --value-basis resident
passes 100000tx_mixed_resident --value-basis resident --workload mixed
passes 125000tx_resident --value-basis resident --txs 125000 --transfers 1
head -12 gpurun_out/${TAG:-run}_kernel_stats_100000tx_t8.csv
