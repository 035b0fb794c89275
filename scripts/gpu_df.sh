set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time THOR_HIP_SPIN_TIMEOUT_S=60 timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "two_streams or 1080p or n6_q32 or ra" ) > gpurun_out/df_tests.log 2>&1 || { tail -15 gpurun_out/df_tests.log; exit 1; }
tail -3 gpurun_out/df_tests.log
( time THOR_HIP_SPIN_TIMEOUT_S=100 timeout 300 python bench.py --no-cpu-baseline --streams 1024 ) > gpurun_out/df_bench1024.log 2>&1; tail -4 gpurun_out/df_bench1024.log
