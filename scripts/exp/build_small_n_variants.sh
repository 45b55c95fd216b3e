#!/bin/bash
# build scripts/exp/_build/small_n_<name> with the product's step TU embedded, one per knob set
#   bash scripts/exp/build_small_n_variants.sh base= one=-DFISHING_X_ONE=3 ...
cd "$(dirname "$0")/../.." || exit 1
mkdir -p scripts/exp/_build
for spec in "$@"; do
  name="${spec%%=*}"; flags="${spec#*=}"
  hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -fno-gpu-rdc -DHARNESS_EMBED_PRODUCT ${flags//,/ } \
      scripts/exp/small_n_shapes.hip -o "scripts/exp/_build/small_n_$name" 2>&1 | grep -E "error" &
done
wait
ls -la scripts/exp/_build
