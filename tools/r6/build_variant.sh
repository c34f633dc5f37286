#!/bin/bash
# one library variant for an experiment: tools/r6/build_variant.sh <name> [-D...]  ->  modelmesh_amd/lib/variants/libmmplace_<name>.so
set -e
cd "$(dirname "$0")/../.."
mkdir -p modelmesh_amd/lib/variants
name=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-unused-function -Wno-unused-variable "$@" \
  modelmesh_amd/csrc/mmplace.hip -o modelmesh_amd/lib/variants/libmmplace_$name.so -ldl -lpthread
echo "built $name"
