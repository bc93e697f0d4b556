#!/bin/bash
# A/B of two builds of the library on one box: tools/ab_two_libs.sh <other.so> <script> [args]   (interleaved twice)
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
OTHER=$1; shift
cp multigrid_dolfinx_amd/libmg_hip.so /tmp/mg_new.so
cp "$OTHER" /tmp/mg_old.so
for round in 1 2; do
  for which in new old; do
    cp /tmp/mg_$which.so multigrid_dolfinx_amd/libmg_hip.so
    echo "== $which (round $round)"
    python "$@" 2>&1 | grep -v "^{" | tail -6
  done
done
cp /tmp/mg_new.so multigrid_dolfinx_amd/libmg_hip.so
