#!/bin/bash
# A/B of a compile-time switch of one kernel file ON THE GPU BOX: bench with the library as built, rebuild <file> with -D<define>, bench again.
#   bash tools/ab_define.sh conv2d_mfma C2M_PRIO=0
set -u
F=$1; D=$2
cd "$GRAFT_REPO_ROOT"
echo "== as built"; bash tools/ab_repeat.sh | cut -c1-70
cd percivaltts_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function -D$D -c $F.hip -o ../../build/csrc/$F.o || exit 1
make > /dev/null 2>&1 || exit 1
cd ../..
echo "== with -D$D"; bash tools/ab_repeat.sh | cut -c1-70
