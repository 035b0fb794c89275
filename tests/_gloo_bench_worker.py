"""Worker of tests/test_dist_gloo.py::test_bench_main_world_size_2: bench.main() end to end under torch.distributed (gloo, no GPU) with the
encoder object replaced by the stand-in that encodes through the host simulation of the engine (tests/test_bench_contract.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import thor_amd
import bench
from test_bench_contract import _HostsimEncoder
thor_amd.Encoder = _HostsimEncoder
sys.argv = ['bench.py', '--gpus', '2', '--streams', '2', '--width', '192', '--height', '128', '--steps', '2', '--warmup', '1'] + os.environ.get('THOR_TEST_BENCH_ARGS', '').split()
bench.main()
