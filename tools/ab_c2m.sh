#!/bin/bash
# A/B of a compile-time switch of conv2d_mfma.hip ON THE GPU BOX, same box, alternating: the fused / forward kernel probe with the library
# as built and with -D<define>.   bash tools/ab_c2m.sh C2M_PIN_WF=0
set -u
D=$1
cd "$GRAFT_REPO_ROOT"
mkdir -p build/ab
cd percivaltts_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function -D$D -c conv2d_mfma.hip -o ../../build/ab/conv2d_mfma.o 2>/dev/null || exit 1
OBJS=$(ls ../../build/csrc/*.o | grep -v conv2d_mfma.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS ../../build/ab/conv2d_mfma.o -o ../../build/ab/libpercival_hip_ab.so || exit 1
cd ../..
for i in 1 2; do
  echo "== as built ($i)"; python tools/c2m_ab_probe.py
  echo "== -D$D ($i)"; PTTS_LIB_PATH=$PWD/build/ab/libpercival_hip_ab.so python tools/c2m_ab_probe.py
done
