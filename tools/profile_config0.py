"""cProfile of one generate() of configs[0] (opt-125m, policy 1/1, B = 1, 32 in / 8 out): where a 4 ms token goes on the host side."""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "isca-2025-lia_amd"))
import torch  # noqa: E402
from lia_amd import hostinfo  # noqa: E402
from lia_amd.generation import generate  # noqa: E402
from lia_amd.model import LiaOPTModel, resolve_shape  # noqa: E402

hostinfo.pin_to_node(hostinfo.gpu_numa_node(0) or 0)
shape = resolve_shape("facebook/opt-125m")
model = LiaOPTModel.random_init(shape, seed=0, n_gpu_layers=0, pin_weight=False)
ids = torch.randint(0, shape.vocab, (1, 32))
kw = dict(max_new_tokens=int(sys.argv[1]) if len(sys.argv) > 1 else 64, min_new_tokens=int(sys.argv[1]) if len(sys.argv) > 1 else 64,
          prefill_policy=1, decoding_policy=1, gpu_percentage=0, pin_weight=False, token_latency=True)
generate(model, ids, **kw)
pr = cProfile.Profile()
pr.enable()
out, lat = generate(model, ids, **kw)
pr.disable()
print("ms per token:", 1e3 * sum(lat[1:]) / len(lat[1:]), "| per step:", [round(1e3 * t, 2) for t in lat[:12]])
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
