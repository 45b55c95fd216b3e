#!/bin/bash
# Run ON THE GPU BOX (through gpurun): three rocprofv3 passes over one bench command.
#   pass 1: --kernel-trace --stats      (per-kernel durations)
#   pass 2: --kernel-trace --pmc FETCH_SIZE     (separate passes: TCC has 4 slots, FETCH_SIZE
#   pass 3: --kernel-trace --pmc WRITE_SIZE      costs 3 and WRITE_SIZE 2 -- MI355X_MICROARCH.md)
# Raw CSVs land under gpurun_out/<tag>/ ; scripts/summarize_profile.py turns them into profiles/.
#   bash scripts/profile_bench.sh <tag> [bench.py flags, e.g. --config v4 --n-envs 16777216]
# The program after `--` is python3 itself (no env / bash -c hop: the profiler's library initialises the GPU).
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
TAG="${1:-prof}"
shift || true
EXTRA="$*"
OUT="$REPO/gpurun_out/$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
COMMON="--no-cpu-baseline --no-subrecords $EXTRA"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- \
    python3 "$REPO/bench.py" --steps 303 --warmup 50 $COMMON > "$OUT/bench_trace.json" 2> "$OUT/trace.err" || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- \
    python3 "$REPO/bench.py" --steps 101 --warmup 20 --spinup-ms 20 $COMMON > /dev/null 2> "$OUT/pmc_fetch.err" || exit 2
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- \
    python3 "$REPO/bench.py" --steps 101 --warmup 20 --spinup-ms 20 $COMMON > /dev/null 2> "$OUT/pmc_write.err" || exit 3
echo "profile passes done: $OUT"
