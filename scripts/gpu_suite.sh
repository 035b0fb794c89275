set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1100 python -m pytest tests -m gpu -q --durations=8 ) > gpurun_out/suite.log 2>&1; tail -30 gpurun_out/suite.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
( time timeout 600 python bench.py ) > gpurun_out/bench.log 2>&1; tail -4 gpurun_out/bench.log
