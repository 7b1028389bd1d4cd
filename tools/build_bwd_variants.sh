#!/bin/bash
# Builds of the library that differ in rasterize_bwd_mm.hip's MI3DGS_BWD_V only (tools/raster_ab.py --libs tools/ab/libmi3dgs_v*.so)
set -e
cd "$(dirname "$0")/../pipeline-pointcloud_amd/csrc"
make -j8 >/dev/null
mkdir -p ../../tools/ab
for v in "$@"; do
  SK=0; case $v in *s*) SK=${v#*s};; esac; VV=${v%%s*}
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form=1 \
     -fno-honor-nans -fno-slp-vectorize -DMI3DGS_BWD_V=$VV -DMI3DGS_BWD_SKIP=$SK -c rasterize_bwd_mm.hip -o ../../tools/ab/bwd_v$v.o &
done
wait
for v in "$@"; do
  OBJS=$(ls build/*.o | grep -v rasterize_bwd_mm.o)
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../tools/ab/libmi3dgs_v$v.so $OBJS ../../tools/ab/bwd_v$v.o
done
ls -la ../../tools/ab/*.so
