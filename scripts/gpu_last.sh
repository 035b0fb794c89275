cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
THOR_HIP_SPIN_TIMEOUT_S=30 timeout 58 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "1080p or n6_q32 or two_streams or q30_ra_gop4 or n4_q32_10bit" > gpurun_out/last2_tests.log 2>&1; tail -3 gpurun_out/last2_tests.log
