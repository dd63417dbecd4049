set -x
O=$GRAFT_REPO_ROOT/gpurun_out/r6e
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  rm -rf /tmp/pe_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pe_$v -- python3 $GRAFT_REPO_ROOT/bench.py --passes timed --cpu-sample 0 --sub-steps 0 --steps 10 --tune 12=$v > $O/line_exp$v.json 2> $O/err_exp$v.txt
  cp $(find /tmp/pe_$v -name '*kernel_stats.csv' | head -1) $O/kernel_stats_exp$v.csv
  grep -E "k_build|k_trace|k_cross|k_finalize" $O/kernel_stats_exp$v.csv | cut -c1-60,200-400 | sed 's/(.*)//' 
done
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "exp_of or K_train or trace or golden" > $O/gputest_some.txt 2>&1; tail -3 $O/gputest_some.txt
