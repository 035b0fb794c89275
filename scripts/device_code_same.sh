#!/bin/bash
# device_code_same.sh COMMIT [FILE]  -  is the gfx950 DEVICE code of thor_amd/csrc/FILE (default thor_hip.cpp) in the working tree the same as at COMMIT?
# Compiles both states with the product's flags to assembly (hipcc --cuda-device-only -S) and diffs them, leaving out the compilation-unit id symbol
# (__hip_cuid_<hash of the source text>) and comments.  Used when a host-side or comment-only change follows a profiled state: profiles/r06_pmc_bench.json
# lists the later source digest under `same_device_code` with this check as its evidence, and bench.py attaches the counters to lines of either state.
set -e
R="$(cd "$(dirname "$0")/.." && pwd)"
c=$1; f=${2:-thor_hip.cpp}
T=$(mktemp -d)
git -C "$R" archive "$c" thor_amd/csrc include | tar -x -C "$T"
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-strict-aliasing -fPIC -pthread --cuda-device-only -S"
/opt/rocm/bin/hipcc $F -o "$T/old.s" "$T/thor_amd/csrc/$f" 2>/dev/null &
/opt/rocm/bin/hipcc $F -o "$T/new.s" "$R/thor_amd/csrc/$f" 2>/dev/null
wait
strip() { grep -v '__hip_cuid_' "$1" | grep -v '^\s*;' | grep -v '^\s*\.file'; }
if diff <(strip "$T/old.s") <(strip "$T/new.s") > "$T/d.txt"; then echo "device code of $f: working tree == $c ($(wc -l < "$T/new.s") lines of assembly)"; rc=0
else echo "device code of $f DIFFERS from $c:"; head -20 "$T/d.txt"; rc=1; fi
rm -rf "$T"; exit $rc
