import sys, tempfile
sys.path.insert(0, "isca-2025-lia_amd")
import torch
from lia_amd import hostinfo
from lia_amd.model import OPTShape
from lia_amd.packed_checkpoint import write_dummy_checkpoint, load_packed
sh = OPTShape("t", 2048, 16, 8192, 6, vocab=4096, max_pos=128)
d = tempfile.mkdtemp(dir="/tmp")
m0 = hostinfo.cgroup_memory()["current"]
write_dummy_checkpoint(sh, d, wire=10)
m1 = hostinfo.cgroup_memory()["current"]
model = load_packed(d, n_gpu_layers=2)
m2 = hostinfo.cgroup_memory()["current"]
print("map modes:", [getattr(st, "map_mode", None) for st in model.layers], "tiers", [st.tier for st in model.layers])
print("cgroup MiB: before write %.0f, after write %.0f, after load+register %.0f; streamed bytes %.0f MiB" % (m0 / 2**20, m1 / 2**20, m2 / 2**20, sum(st.stream_bytes for st in model.layers[2:]) / 2**20))
