"""Developer A/B: run bench workloads against an alternative build of the library (PDEGYM_LIB=path)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pdecontrolgym_amd import _native as N
if os.environ.get("PDEGYM_LIB"):
    N.LIB_PATH = os.environ["PDEGYM_LIB"]
import bench
sys.argv = ["bench.py"] + sys.argv[1:]
bench.main()
