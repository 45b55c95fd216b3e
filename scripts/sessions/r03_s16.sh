#!/bin/bash
# round 3, session 16: SQ-counter passes at the launch-bound sizes (how long do the waves live inside the kernel's duration?)
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
O="$REPO/gpurun_out/r03_s16"; mkdir -p "$O"
cd "$REPO"
for spec in "v2_2p19:--config v2 --n-envs 524288" "v1_2p20:--n-envs 1048576" "v1_2p22:"; do
  tag="${spec%%:*}"; extra="${spec#*:}"
  bash scripts/profile_sq.sh "r03_s16/sq_$tag" $extra > /dev/null || { echo "sq $tag failed"; exit 2; }
  python3 scripts/summarize_sq.py "$O/sq_$tag" "$O/r03_sq_$tag.json" > "$O/r03_sq_$tag.txt" || exit 3
  rm -rf "$O/sq_$tag/pmc_sq"
done
echo done
