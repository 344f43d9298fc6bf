import sys, os, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + "/isca-2025-lia_amd", R + "/tests/golden", R + "/oracle"]
import torch, synth, lia_oracle as o
from lia_amd import ops
ctx = ops.Context(0, 1 << 30)
def dev(b): return torch.from_numpy(np.ascontiguousarray(b).view(np.int16)).view(torch.bfloat16).cuda()
def bits(t): return t.cpu().contiguous().view(torch.int16).numpy().view(np.uint16)
def rb(seed, shape, s=1.0): return synth.f32_to_bf16_bits((s*np.random.RandomState(seed).standard_normal(shape)).astype(np.float32))
for (M,N,K,split) in [(64,512,1024,1),(64,512,1024,2),(16,512,1024,2),(32,512,1024,1),(64,64,128,1),(64,64,256,1),(64,64,512,1),(48,64,256,1)]:
    x, w = rb(10,(M,K)), rb(11,(N,K),K**-0.5)
    y = ctx.linear(dev(x), dev(w), split_k=split); ctx.synchronize()
    ref = o.linear(x, w)
    e = np.abs(synth.bf16_bits_to_f32(bits(y)) - synth.bf16_bits_to_f32(ref)) > 0.05
    print(M,N,K,split,'bad',e.sum(),'rows',np.unique(np.where(e)[0])[:20],'cols',np.unique(np.where(e)[1])[:24])
