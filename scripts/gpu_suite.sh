set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time python -m pytest tests -m gpu -x -q ) 2>&1 | tail -15
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
( time python bench.py --steps 2 --warmup 1 --streams 128 ) 2>&1 | tail -6
