#!/bin/bash
# usage: bash tools/build_variant.sh <name> [-DFLAG=... ...]   ->  build_var/<name>.so  (A/B builds of the SAME library; tools/ab_so.py)
set -e
cd "$(dirname "$0")/.."
mkdir -p build_var
name=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -pthread "$@" -o build_var/$name.so mirge3.0_amd/csrc/mirge_native.hip -lz
echo build_var/$name.so
