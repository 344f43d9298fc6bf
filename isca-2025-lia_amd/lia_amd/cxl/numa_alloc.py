"""numa_alloc_tensor / numa_free_tensor: same names and contract as lia/cxl/numa_alloc.py:28-55, on top of the
four C exports liblia_hip.so shares with the reference's shim (numa_alloc.c).

Differences: the interleave node set comes from LIA_CXL_NODES / set_cxl_nodes() instead of the hard-coded {2,3}
(numa_alloc.c:80-81), and the range can be registered with the GPU driver (register=True) so the copy engine DMAs
straight from it -- the reference leaves it pageable (`# return tensor.pin_memory()` is commented out,
numa_alloc.py:49), which turns every copy_(non_blocking=True) into a staged synchronous copy.
"""
import ctypes

import numpy as np
import torch

from .. import _native as N

_registered = {}


def set_cxl_nodes(nodes):
    arr = (ctypes.c_int * len(nodes))(*nodes)
    N.check(N.lib().lia_numa_set_interleave_nodes(arr, len(nodes)), "lia_numa_set_interleave_nodes")


def numa_alloc_tensor(shape, dtype, register=False):
    element_size = torch.tensor([], dtype=dtype).element_size()
    total_size = int(np.prod(shape)) * element_size
    ptr = N.lib().numa_alloc_interleave(total_size)
    if not ptr:
        print("Memory allocation failed")          # numa_alloc.py:33-35: message + None
        return None
    if register:
        rc = N.lib().lia_numa_register(ptr, total_size)
        if rc != 0:
            N.lib().numa_free_node(ptr, total_size)
            N.check(rc, "lia_numa_register")
        _registered[ptr] = total_size
    buffer = (ctypes.c_char * total_size).from_address(ptr)
    t = torch.frombuffer(buffer, dtype=torch.uint8).view(dtype).reshape(shape)   # zero-copy view of the NUMA range
    return t


def numa_free_tensor(tensor):
    ptr = tensor.data_ptr()
    size = tensor.nelement() * tensor.element_size()
    if ptr in _registered:
        N.lib().lia_numa_unregister(ptr)
        del _registered[ptr]
    N.lib().numa_free_node(ptr, ctypes.c_size_t(size))
