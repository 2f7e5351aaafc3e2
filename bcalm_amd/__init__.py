"""bcalm_amd -- MI355X-native compacted de Bruijn graph construction (reads -> unitigs).

The product is the HIP shared library bcalm_amd/_build/libcdbg.so (C ABI in
include/cdbg.h) plus the `bcalm` command-line host; this package is the thin Python
mirror used by tests and bench.py.  No CPU fallback exists anywhere in this package.
"""
from .api import CdbgError, Graph, load, DEFAULT_LIB, EXPORTS  # noqa: F401
