#!/bin/bash
# A/B of compile-time switches of ONE kernel file ON THE GPU BOX, same box: a probe with the library as built and with each -D<define>.
#   bash tools/ab_file.sh dense tools/dense_ramp_probe.py DNS_DEPTH=1 DNS_DEPTH=2
set -u
F=$1; P=$2; shift 2
cd "$GRAFT_REPO_ROOT"
mkdir -p build/ab
echo "== as built"; python $P 2>&1 | grep -v amdgpu.ids
for D in "$@"; do
  (cd percivaltts_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function -D$D -c $F.hip -o ../../build/ab/$F.o 2>/dev/null &&
   /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $(ls ../../build/csrc/*.o | grep -v "/$F.o") ../../build/ab/$F.o -o ../../build/ab/libpercival_hip_ab.so) || exit 1
  echo "== -D$D"; PTTS_LIB_PATH=$PWD/build/ab/libpercival_hip_ab.so python $P 2>&1 | grep -v amdgpu.ids
done
echo "== as built (again)"; python $P 2>&1 | grep -v amdgpu.ids
