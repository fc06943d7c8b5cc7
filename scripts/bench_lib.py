"""bench.py against another build of the engine: python scripts/bench_lib.py said_amd/lib/ab_<name>.so [bench flags] (same-box A/Bs of -D variants, scripts/build_variant.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from said_amd import _engine
_engine._LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = ["bench.py"] + sys.argv[2:]
import bench
bench.main()
