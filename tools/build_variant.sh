#!/bin/bash
# Builds a variant of libsv3d_hip.so under gpurun_variants/ (git-ignored, shipped to the GPU box):
#   tools/build_variant.sh NAME [extra hipcc flags, e.g. -DSV3D_TRACE] [RASTER=path/to/raster.hip]
set -e
name=$1; shift
src=single-view-3d-reconstruction_amd/csrc
raster=$src/raster.hip
flags=()
for a in "$@"; do
  case $a in RASTER=*) raster=${a#RASTER=};; *) flags+=("$a");; esac
done
mkdir -p gpurun_variants/obj
make -s -C $src >/dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fno-slp-vectorize --offload-arch=gfx950 -Iinclude -I$src -Wall -Wno-unused-function \
  -mllvm -amdgpu-kernarg-preload-count=16 "${flags[@]}" -c $raster -o gpurun_variants/obj/$name.raster.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o gpurun_variants/$name.so gpurun_variants/obj/$name.raster.o \
  $src/build/chamfer.hip.o $src/build/mesh.hip.o $src/build/meshreg.hip.o $src/build/texture.hip.o $src/build/abi.cpp.o
echo gpurun_variants/$name.so
