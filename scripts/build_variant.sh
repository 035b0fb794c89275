#!/bin/bash
# build_variant.sh NAME [extra hipcc flags...]  ->  thor_amd/libthor_hip_NAME.so (A/B variants of the same sources; the
# product library thor_amd/libthor_hip.so is built by __graft_entry__.build()).  THOR_HIP_LIB=<path> selects a variant.
set -e
R="$(cd "$(dirname "$0")/.." && pwd)"
name=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-strict-aliasing -fPIC -shared -pthread "$@" \
  -o "$R/thor_amd/libthor_hip_$name.so" "$R/thor_amd/csrc/thor_hip.cpp" $( [ -f "$R/thor_amd/csrc/thor_hip_lat.cpp" ] && echo "$R/thor_amd/csrc/thor_hip_lat.cpp" ) $( [ -f "$R/thor_amd/csrc/thor_hip_wide.cpp" ] && echo "$R/thor_amd/csrc/thor_hip_wide.cpp" ) 2>&1 | grep -E "error:" || true
ls -la "$R/thor_amd/libthor_hip_$name.so"
