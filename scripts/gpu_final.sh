set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/rocprof /tmp/w
export TMPDIR=/tmp
( time THOR_HIP_SPIN_TIMEOUT_S=120 timeout 330 python -m pytest tests -m gpu -q --durations=6 ) > gpurun_out/final_tests.log 2>&1; tail -14 gpurun_out/final_tests.log
( time timeout 200 rocprofv3 --kernel-trace --stats -d gpurun_out/rocprof -o r01final -- python bench.py ) > gpurun_out/final_bench_rocprof.log 2>&1; grep '"metric"' gpurun_out/final_bench_rocprof.log | cut -c1-900
ls -la gpurun_out/rocprof/* | head
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python3 -m thor_amd.synth /tmp/w/hd.yuv 1920 1080 6 2
THOR_PROF=1 THOR_HIP_SPIN_TIMEOUT_S=100 timeout 120 tools/thorenc_hip_prof -cf configs/ldb_high_efficiency.cfg -if /tmp/w/hd.yuv -width 1920 -height 1080 -qp 32 -f 30 -n 3 -streams 256 -wrap 4 > gpurun_out/prof4_1080p_s256.log 2>&1; head -3 gpurun_out/prof4_1080p_s256.log
