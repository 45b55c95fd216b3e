#!/bin/bash
# round 5, session 26: SQ-counter passes (wave-cycle shares, VALU issue utilisation) of the step kernels at the judged sizes on the
# round's final build -- the metric (fishing-v1, N = 2^22), config 5's shard (fishing-v4, 2^21), config 4's shard (fishing-v2, 2^19),
# config 2 (fishing-v1, 2^20) -- next to round 4's (profiles/r04_sq/)
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
O="$REPO/gpurun_out/r05_s26"; mkdir -p "$O/summ"
cd "$REPO"
for spec in "v1_2p22:--n-envs 4194304" "v4_2p21:--config v4" "v2_2p19:--config v2 --n-envs 524288" "v1_2p20:--n-envs 1048576"; do
  tag="${spec%%:*}"; extra="${spec#*:}"
  bash scripts/profile_sq.sh "r05_s26/sq_$tag" $extra > /dev/null || { echo "sq $tag failed"; tail -5 "$O/sq_$tag/sq.err"; exit 2; }
  python3 scripts/summarize_sq.py "$O/sq_$tag" "$O/summ/r05_sq_$tag.json" > /dev/null || { echo "summary $tag failed"; exit 3; }
  rm -rf "$O/sq_$tag/pmc_sq"
  echo "sq $tag done"
done
echo done
