#!/bin/bash
# Regenerate everything under profiles/ for one round on the GPU box:  tools/make_profiles.sh r03
# Outputs land in gpurun_out/profiles_<tag>/ (copy the ones to keep into profiles/).
tag=${1:-r03}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/profiles_$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
set -x
python3 $R/bench.py > $O/${tag}_bench_line.json 2> $O/bench.err || exit 1
for ov in 1 0; do
  name=$([ $ov = 1 ] && echo bench || echo bench_overlap0)
  rm -rf /tmp/prof_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -- python3 $R/bench.py --overlap $ov --cpu-sample 0 --sub-steps 0 > $O/${tag}_${name}_line_under_rocprof.json 2> $O/$name.err || exit 1
  cp $(find /tmp/prof_$name -name '*kernel_stats.csv' | head -1) $O/${tag}_${name}_n8192_kernel_stats.csv
done
python3 $R/bench.py --rows 1500 --experts-total 16 --cpu-sample 0 > $O/${tag}_bench_bcm16_line.json 2>> $O/bench.err
python3 $R/bench.py --rows 6000 --experts-total 4 --cpu-sample 0 --steps 10 > $O/${tag}_bench_bcm4_line.json 2>> $O/bench.err
python3 $R/bench.py --experts-per-gpu 2 --cpu-sample 0 --steps 10 > $O/${tag}_bench_2x8192_line.json 2>> $O/bench.err
python3 $R/tools/la_bench.py > $O/${tag}_la_bench.txt 2>> $O/bench.err
python3 $R/tools/bcm_ab.py > $O/${tag}_bcm_one_gpu.txt 2>> $O/bench.err
for mode in overlap serial; do
  rm -rf /tmp/tl_$mode
  pipe=$([ $mode = overlap ] && echo -1 || echo 0)
  rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$mode -- python3 $R/tools/timeline_run.py 8192 $pipe > /dev/null 2>&1 || exit 1
  python3 $R/tools/timeline_report.py $(find /tmp/tl_$mode -name '*kernel_trace.csv' | head -1) --launches > $O/${tag}_timeline_${mode}_n8192.txt
done
# one 1500-row expert and the 16 x 1500 group (config 5's shape on one GPU)
for shape in "expert1500 1500 0" "bcm16x1500 24000 16"; do
  set -- $shape
  rm -rf /tmp/tl_$1
  TL_EXPERTS=$3 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$1 -- python3 $R/tools/timeline_run.py $2 > /dev/null 2>&1 || exit 1
  python3 $R/tools/timeline_report.py $(find /tmp/tl_$1 -name '*kernel_trace.csv' | head -1) --launches > $O/${tag}_timeline_$1.txt
done
# PMC passes (separate passes; counters only with --kernel-trace)
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_LDS_BANK_CONFLICT" "FETCH_SIZE" "WRITE_SIZE"; do
  t=$(echo $pass | cut -d' ' -f1)
  rm -rf $R/gpurun_out/pmc_$t
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$t -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 --sub-steps 0 > $R/gpurun_out/pmc_$t.log 2>&1 || echo "pass $t failed"
done
cd $R && python3 tools/pmc_summary.py $R/gpurun_out $tag && cp profiles/${tag}_pmc_summary.json $O/
# the factorisation's chain kernels alone, with in-kernel cycle stamps (built in the build container: see the sources' headers)
[ -x $R/tools/bin/chain_bench ] && $R/tools/bin/chain_bench > $O/${tag}_chain_bench.txt 2>&1
[ -x $R/tools/bin/panel_bench ] && $R/tools/bin/panel_bench >> $O/${tag}_chain_bench.txt 2>&1
ls -la $O
