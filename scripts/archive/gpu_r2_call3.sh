#!/bin/bash
# round 2, call 3: streams-vs-throughput of the 4-wave engine in the 4-reference regime (P frames 4..5), against the same
# sources built with 1 and 2 waves per workgroup and with -DTHOR_EXP_UNIFORM; rocprofv3 kernel stats of the default build
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
run() {  # tag lib streams
  THOR_HIP_LIB=$PWD/thor_amd/$2 timeout 900 python bench.py --streams $3 --warmup 4 --steps 2 --no-verify --no-cpu-baseline > gpurun_out/r2c3_$1_s$3.log 2>&1
  echo "$1 S=$3: $(grep -o '"value": [0-9.]*' gpurun_out/r2c3_$1_s$3.log | head -1) $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r2c3_$1_s$3.log)"
}
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2c3_rocprof -o w4_s96 -- python $GRAFT_REPO_ROOT/bench.py --streams 96 --warmup 4 --steps 2 --no-verify --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r2c3_w4_s96.log 2>&1; cd $GRAFT_REPO_ROOT
echo "w4 S=96 (rocprofv3): $(grep -o '"value": [0-9.]*' gpurun_out/r2c3_w4_s96.log | head -1) $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r2c3_w4_s96.log)"
run w4 libthor_hip.so 48
run w4 libthor_hip.so 192
run uni libthor_hip_uni.so 96
run w1 libthor_hip_w1.so 96
run w1 libthor_hip_w1.so 192
run w2 libthor_hip_w2.so 96
find gpurun_out/r2c3_rocprof -name "*stats*" | head; ls -la gpurun_out/r2c3_rocprof | head
