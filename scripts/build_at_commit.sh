#!/bin/bash
# build_at_commit.sh COMMIT NAME  ->  thor_amd/libthor_hip_NAME.so built from the engine sources of COMMIT (A/B of committed states
# on the GPU box: built libraries travel with the snapshot; THOR_HIP_LIB=<path> selects one).  Run in the build container.
set -e
R="$(cd "$(dirname "$0")/.." && pwd)"
c=$1; name=$2
T=$(mktemp -d)
git -C "$R" archive "$c" thor_amd/csrc include | tar -x -C "$T"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-strict-aliasing -fPIC -shared -pthread \
  -o "$R/thor_amd/libthor_hip_$name.so" "$T/thor_amd/csrc/thor_hip.cpp" $( [ -f "$T/thor_amd/csrc/thor_hip_lat.cpp" ] && echo "$T/thor_amd/csrc/thor_hip_lat.cpp" ) $( [ -f "$T/thor_amd/csrc/thor_hip_wide.cpp" ] && echo "$T/thor_amd/csrc/thor_hip_wide.cpp" ) 2>&1 | grep -E "error:" || true
rm -rf "$T"
ls -la "$R/thor_amd/libthor_hip_$name.so"
