#!/bin/bash
# Build container: timing-only variants of the matrix-core last conv (results are wrong by construction) into tools/ablate_builds/:
#   nomfma  copies + barriers + epilogue, no LDS reads / MFMAs        nodma  arithmetic only (two stages copied once)
# The GPU job (tools/sessions/gpu_r3_s20.sh) swaps each over the library of its scratch copy and times the class.
set -e
cd "$(dirname "$0")/../livespeechportraits_amd/csrc"
mkdir -p ../../tools/ablate_builds
OTHERS=$(ls build/*.o | grep -v "edge_layers.o\|-hip-amdgcn")
for v in NOMFMA NODMA; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DLC_ABL_$v -DLC_$v -c edge_layers.hip -o /tmp/edge_$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/ablate_builds/liblspf2f_$v.so $OTHERS /tmp/edge_$v.o
done
# the rasteriser with phase stamps (tools/raster_stamps.py)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DLSPRASTER_STAMPS -c raster.hip -o /tmp/raster_stamps.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/ablate_builds/liblspf2f_RSTAMPS.so $(ls build/*.o | grep -v "raster.o\|-hip-amdgcn") /tmp/raster_stamps.o
ls -la ../../tools/ablate_builds
