#!/bin/bash
# several -D sets of conv2d_mfma.hip against the library as built, same box:   bash tools/ab_c2m_multi.sh probe.py "A=1" "A=2 B=1" ...
set -u
P=$1; shift
cd "$GRAFT_REPO_ROOT"
mkdir -p build/ab
echo "== as built"; python $P 2>&1 | grep -v amdgpu.ids
for DS in "$@"; do
  FL=""; for D in $DS; do FL="$FL -D$D"; done
  (cd percivaltts_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function $FL -c conv2d_mfma.hip -o ../../build/ab/conv2d_mfma.o 2>/dev/null &&
   /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $(ls ../../build/csrc/*.o | grep -v "/conv2d_mfma.o") ../../build/ab/conv2d_mfma.o -o ../../build/ab/libpercival_hip_ab.so) || exit 1
  echo "== $DS"; PTTS_LIB_PATH=$PWD/build/ab/libpercival_hip_ab.so python $P 2>&1 | grep -v amdgpu.ids
done
