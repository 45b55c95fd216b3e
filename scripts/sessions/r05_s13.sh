#!/bin/bash
# round 5, session 13: the zoo's wave-uniform arguments pinned in SGPRs (FISHING_LEAN_PIN_ZOO_ARGS: LLVM re-loaded them from the kernarg
# segment in every per-env block -- ten s_load + s_waitcnt lgkmcnt(0) round trips strung through fishing-v11's arithmetic) and
# fishing-v11's four lookup scalars fetched ahead of the table's barrier.  pinzoo = the source as it stood during the session,
# nopinzoo = the same with -DFISHING_LEAN_PIN_ZOO_ARGS=0.  Verdict: not adopted (profiles/r05_pin_zoo_args.jsonl); the knob left the
# tree with it -- the kernel's comment behind its argument batch says what it did
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
O="$REPO/gpurun_out/r05_s13"; mkdir -p "$O"
cd "$REPO"
FISHING_HIP_LIB="$REPO/gym_fishing_amd/_lib/variants/libfishing_hip_pinzoo.so" timeout -k 10 600 python3 -m pytest tests/test_gpu_zoo.py tests/test_gpu_fused_and_dispatch.py tests/test_gpu_envs.py tests/test_gpu_simulate.py -m gpu -q -x > "$O/tests_pinzoo.log" 2>&1 || { tail -30 "$O/tests_pinzoo.log" | cut -c1-250; exit 1; }
tail -1 "$O/tests_pinzoo.log"
for rep in 1 2 3; do
  for var in nopinzoo pinzoo; do
    FISHING_HIP_LIB="$REPO/gym_fishing_amd/_lib/variants/libfishing_hip_$var.so" timeout -k 10 300 python3 scripts/exp/run_f_rows.py --zoo-only > "$O/rows_${var}_$rep.jsonl" 2> "$O/rows.err" || { echo "$var rows failed"; tail -5 "$O/rows.err"; exit 3; }
  done
done
echo done
