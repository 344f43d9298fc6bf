"""cProfile of the host side of a few decode steps (where the time between two steps goes): python tools/pyprof_decode.py [llama|opt]"""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "isca-2025-lia_amd"))
import torch
from lia_amd.generation import generate
from lia_amd import hostinfo
which = sys.argv[1] if len(sys.argv) > 1 else "llama"
node = hostinfo.gpu_numa_node(0)
print("pinned cpus:", hostinfo.pin_to_node(node) if node >= 0 else 0, "torch threads (default):", torch.get_num_threads())
if len(sys.argv) > 2:
    torch.set_num_threads(int(sys.argv[2]))
    print("torch threads set to", torch.get_num_threads())
if which == "llama":
    from lia_amd.llama import LiaLlamaModel, resolve_llama_shape
    model = LiaLlamaModel.random_init(resolve_llama_shape("llama-3-8b"), seed=0)
    B, T, flags = 128, 1024, dict(gpu_percentage=100, pin_weight=True)
else:
    from lia_amd.model import LiaOPTModel, resolve_shape
    model = LiaOPTModel.random_init(resolve_shape("opt-30b"), seed=0, n_gpu_layers=48)
    B, T, flags = 64, 256, dict(gpu_percentage=100, pin_weight=True, prefill_policy=0, decoding_policy=2)
g = torch.Generator().manual_seed(0)
row = torch.randint(4, 30000, (T,), generator=g)
ids = row[None, :].repeat(B, 1)
generate(model, ids, max_new_tokens=4, min_new_tokens=4, do_sample=False, num_beams=1, token_latency=True, **flags)
pr = cProfile.Profile()
state = {}
def hook(step):
    if step == 2:
        pr.enable()
out, lat = generate(model, ids, max_new_tokens=34, min_new_tokens=34, do_sample=False, num_beams=1, token_latency=True, step_hook=hook, **flags)
pr.disable()
print("decode ms/step:", 1e3 * sum(lat[2:]) / len(lat[2:]))
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
print(s.getvalue()[:6000])
