#!/bin/bash
# kernel timeline of one LL+grad evaluation:  tools/timeline_eval.sh <tag> <n> [pipe [key=value ...]]  -> gpurun_out/tl_<tag>.txt
tag=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl_$tag
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$tag -- python3 $R/tools/timeline_run.py "$@" > $R/gpurun_out/tl_${tag}_run.log 2>&1 || { echo "rocprofv3 failed"; tail -5 $R/gpurun_out/tl_${tag}_run.log; exit 1; }
f=$(find /tmp/tl_$tag -name '*kernel_trace.csv' | head -1)
python3 $R/tools/timeline_report.py $f --launches > $R/gpurun_out/tl_$tag.txt
head -22 $R/gpurun_out/tl_$tag.txt
