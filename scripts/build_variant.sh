#!/bin/bash
# Usage: scripts/build_variant.sh <tag> [extra hipcc flags...]   ->  ohm_amd/lib/variants/libohmhip_<tag>.so
# Development A/B builds of the library (e.g. -DOHMHIP_WALK_UNROLL=4); select one at run time with OHMHIP_LIB=<path>.
set -e
cd "$(dirname "$0")/.."
TAG=$1; shift
mkdir -p ohm_amd/lib/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wall -Wno-unused-function "$@" \
  -o ohm_amd/lib/variants/libohmhip_$TAG.so ohm_amd/csrc/ohmhip_device.hip ohm_amd/csrc/ohmhip_map.hip ohm_amd/csrc/ohmhip_transform.hip -L/opt/rocm/lib -lrccl
echo built ohm_amd/lib/variants/libohmhip_$TAG.so
