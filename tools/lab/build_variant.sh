#!/bin/bash
# An A/B build of the library with extra compiler flags (-DSPLAT_...=...), next to the product:
#   bash tools/lab/build_variant.sh name [flags...]   ->  splat_amd/ab/libsplat_<name>.so
# (git-ignored, but it travels to the GPU box with the snapshot; run it there with SPLAT_AMD_LIB=... or tools/lab/kern_ab.py)
set -e
name=$1; shift
cd "$(dirname "$0")/../../splat_amd/csrc"
mkdir -p ../ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-result -Wno-bitwise-instead-of-logical \
  -fno-slp-vectorize "$@" -shared -o ../ab/libsplat_$name.so splat_api.hip splat_kernels.hip splat_multi.hip splat_policy.cpp -ldl -lpthread
ls -la ../ab/libsplat_$name.so
