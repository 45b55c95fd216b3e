#!/bin/bash
# GPU session 3a of round 2: full GPU suite, rocprofv3 kernel stats + PMC passes for the v1 / v0 / v2 workloads.
set -u
O=gpurun_out/r02_s3
mkdir -p $O
timeout -k 10 700 python -m pytest tests -m gpu -q -p no:cacheprovider --maxfail=20 > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log
for spec in "v1:" "v1_bare:--no-returns" "v0:--config v0" "v2:--config v2"; do
  tag=${spec%%:*}; flags=${spec#*:}
  echo "== profile $tag ($flags)"; bash scripts/profile_bench.sh r02_s3/prof_$tag $flags; echo rc=$?
done
