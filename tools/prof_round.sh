set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -u bench.py > gpurun_out/${TAG:-run}_bench_n1.json 2> gpurun_out/${TAG:-run}_bench_n1.err
tail -c 600 gpurun_out/${TAG:-run}_bench_n1.json
bash tools/prof_kernels.sh gpurun_out/${TAG:-run}_kernel_stats_100000tx_t8.csv --steps 3 --warmup 1 --no-secondary --e2e-reps 0
for c in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU; do
  n=$(echo $c | tr 'A-Z' 'a-z'); [ $c = SQ_INSTS_VALU ] && n=valu
  bash tools/prof_pmc.sh $c gpurun_out/${TAG:-run}_pmc_${n}_100000tx_t8.csv --no-secondary --e2e-reps 0
  echo "pmc $c done"
done
head -30 gpurun_out/${TAG:-run}_kernel_stats_100000tx_t8.csv
