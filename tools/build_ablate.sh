#!/bin/bash
# builds diagnostic variants of the library into tools/abl/ (scratch; not shipped)
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/abl
for a in ${ABL_LIST:-0 1 2 4 7}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I include -I lcaonet_amd/csrc -DLCAO_ABL=$a ${ABL_FLAGS} \
     lcaonet_amd/csrc/lcao_mfma.hip lcaonet_amd/csrc/lcao_generic.hip -o tools/abl/lib$a.so &
done
wait
ls -la tools/abl
