#!/bin/bash
set -u
O=gpurun_out/r02_s4
mkdir -p $O
timeout -k 10 700 python -m pytest tests -m gpu -q -p no:cacheprovider --maxfail=20 > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log
echo "== caps sweep"; timeout -k 10 600 python scripts/exp/sweep_caps_large.py > $O/caps_large.jsonl 2> $O/caps.err; cat $O/caps_large.jsonl
echo "== A/B"; timeout -k 10 600 python scripts/exp/ab_r02.py > $O/ab.jsonl 2> $O/ab.err; cat $O/ab.jsonl
