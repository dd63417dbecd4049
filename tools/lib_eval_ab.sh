#!/bin/bash
# A/B of two library builds on one board, whole evaluations: tools/lib_eval_ab.sh <base .so> <out file> [sizes...]
# (the tree's cugp_amd/lib/libcugp.so is the candidate; single matrices via tools/ab2.py, expert groups via tools/bcm_ab.py)
base=$1; out=$2; shift 2
sizes=${@:-8192}
R=${GRAFT_REPO_ROOT:-/root/repo}
cp $R/cugp_amd/lib/libcugp.so /tmp/cand.so
: > $out
for rnd in 1 2; do
  for which in base cand; do
    if [ $which = base ]; then cp $base $R/cugp_amd/lib/libcugp.so; else cp /tmp/cand.so $R/cugp_amd/lib/libcugp.so; fi
    echo "== $which round $rnd" >> $out
    for n in $sizes; do
      AB_ROUNDS=5 AB_LA=0 python3 $R/tools/ab2.py $n base 2>&1 | grep "eval wall" | cut -c1-90 | sed "s/^base/n=$n/" >> $out
    done
    python3 $R/tools/bcm_ab.py 2>&1 | grep "K=" >> $out
  done
done
cp /tmp/cand.so $R/cugp_amd/lib/libcugp.so
