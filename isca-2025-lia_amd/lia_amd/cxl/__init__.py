"""CXL / NUMA host tier: counterpart of the reference's lia/cxl/ (numa_alloc.py, benchmark.py, run.sh)."""
from .numa_alloc import numa_alloc_tensor, numa_free_tensor  # noqa: F401
