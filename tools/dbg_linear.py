import sys, os, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + "/isca-2025-lia_amd", R + "/tests/golden", R + "/oracle"]
import torch, synth, lia_oracle as o
from lia_amd import ops
ctx = ops.Context(0, 1 << 30)
def dev(b): return torch.from_numpy(np.ascontiguousarray(b).view(np.int16)).view(torch.bfloat16).cuda()
def bits(t): return t.cpu().contiguous().view(torch.int16).numpy().view(np.uint16)
def rb(seed, shape, s=1.0): return synth.f32_to_bf16_bits((s*np.random.RandomState(seed).standard_normal(shape)).astype(np.float32))
shapes = eval(sys.argv[1]) if len(sys.argv) > 1 else [(130,768,512,0),(130,768,512,1),(256,768,512,0),(130,64,128,1),(130,64,256,1),(129,64,512,1),(200,64,512,1)]
for (M,N,K,split) in shapes:
    x, w = rb(10,(M,K)), rb(11,(N,K),K**-0.5)
    y = ctx.linear(dev(x), dev(w), split_k=split); ctx.synchronize()
    ref = o.linear(x, w)
    e = np.abs(synth.bf16_bits_to_f32(bits(y)) - synth.bf16_bits_to_f32(ref)) > 0.05
    print(M,N,K,split,'bad',e.sum(),'rows',np.unique(np.where(e)[0])[:20],'cols',np.unique(np.where(e)[1])[:24])
