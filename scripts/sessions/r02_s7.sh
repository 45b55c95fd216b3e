#!/bin/bash
set -u
O=gpurun_out/r02_s7
mkdir -p $O
timeout -k 10 700 python -m pytest tests -m gpu -q -p no:cacheprovider --maxfail=20 > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
echo "== A/B"; timeout -k 10 600 python scripts/exp/ab_r02.py base > $O/ab.jsonl 2> $O/ab.err; cat $O/ab.jsonl
