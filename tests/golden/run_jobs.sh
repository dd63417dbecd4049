#!/bin/bash
# Run the round-2 reference jobs of make_golden.py, P at a time (default 5), longest first.
# Build-container only (needs /root/reference); logs under /tmp/golden_jobs.
cd "$(dirname "$0")/../.."
P=${1:-5}
mkdir -p /tmp/golden_jobs
python tests/golden/make_golden.py --job data || exit 1
python tests/golden/make_golden.py --job list | \
  xargs -P "$P" -I{} sh -c 'python tests/golden/make_golden.py --job {} > /tmp/golden_jobs/{}.log 2>&1'
