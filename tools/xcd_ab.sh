#!/bin/bash
# A/B of two library builds on one board (whole evaluation via tools/ab2.py, LA ops via tools/lib_ab.py):
#   tools/xcd_ab.sh <base .so> <out dir> [sizes...]      (the tree's cugp_amd/lib/libcugp.so is the candidate)
base=$1; O=$2; shift 2
sizes=${@:-8192}
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $O
cp $R/cugp_amd/lib/libcugp.so /tmp/cand.so
for rnd in 1 2; do
  for which in base cand; do
    if [ $which = base ]; then cp $base $R/cugp_amd/lib/libcugp.so; else cp /tmp/cand.so $R/cugp_amd/lib/libcugp.so; fi
    for n in $sizes; do
      echo "== $which n=$n round $rnd" >> $O/ab.txt
      AB_ROUNDS=5 python3 $R/tools/ab2.py $n base >> $O/ab.txt 2>&1 || exit 1
    done
  done
done
cp /tmp/cand.so $R/cugp_amd/lib/libcugp.so
python3 $R/tools/lib_ab.py $base /tmp/cand.so 4096 8192 16384 > $O/lib_ab.txt 2>&1
