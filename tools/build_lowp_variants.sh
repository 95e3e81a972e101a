#!/bin/bash
# builds variants of lcao_lowp.hip into tools/abl/lowp_<tag>.so (scratch; not shipped).  VARIANTS="tag:flags;tag:flags"
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/abl
IFS=';' read -ra VS <<< "${VARIANTS:-cur:}"
for v in "${VS[@]}"; do
  tag="${v%%:*}"; fl="${v#*:}"
  src="${SRC:-lcaonet_amd/csrc/lcao_lowp.hip}"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I include -I lcaonet_amd/csrc $fl "$src" -o tools/abl/lowp_$tag.so &
done
wait
ls -la tools/abl
