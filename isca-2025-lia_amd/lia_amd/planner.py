"""Analytical planner: pick gpu% / policies / num-minibatch for a model, batch and box.

The reference hand-picks these per script line (llm/scripts/lia_{online,offline}.sh) and the paper's cost model is not in
the tree; this is a small roofline model of THIS implementation, calibrated on the r01 measurements:

  per streamed layer    copy      = layer_bytes / link_gbs                       (57 GB/s pinned H2D on the box)
  decode, resident      gpu       = layer_bytes / hbm_gbs + kv_bytes / hbm_gbs   (skinny GEMM ~4.8 TB/s, attention ~6 TB/s)
  decode, policy 2      host attn = kv_bytes / min(threads * 23 GB/s, 220 GB/s)  (row-run prefetch; 13 GB/s per core before)
  prefill               gemm      = flops / mfma_tflops                          (1.15 PFLOP/s measured)
  a streamed layer costs max(copy, its compute); a forward is the sum over layers (+ a fill bubble in prefill).
"""
from dataclasses import dataclass

from . import hostinfo


@dataclass
class Box:
    link_gbs: float = 57.0            # pinned H2D, measured (PCIe Gen5 x16 spec 63)
    hbm_gb: float = 288.0
    hbm_gbs: float = 4800.0           # what the decode GEMM sustains at M = 64
    attn_gbs: float = 6000.0          # decode attention on HBM-resident KV
    mfma_tflops: float = 1200.0       # prefill GEMM on random data, in the pipeline (GM = 4 tile order)
    host_gbs_per_thread: float = 23.0  # host attention with the row-run prefetch: 46 GB/s at 2 threads ...
    host_gbs_cap: float = 220.0        # ... and 223 GB/s at 16 (the socket's DRAM)
    # host-computed layers, threads pinned to the NUMA node that holds the weights (bench.py / run_generation.py do that):
    host_linear_gbs_per_thread: float = 4.4   # policy-1 linears at M = 64 beside the running weight stream (4.9 alone; 3.6 unpinned)
    host_attn_beside_stream: float = 0.8      # share of its rate the host attention keeps beside the stream (0.55 unpinned)
    wire_ratio: float = 0.675         # bytes shipped per weight byte: pack10 0.675, pack11 0.696, pack12 0.751, raw 1.0
    host_threads: int = 0
    host_mem_gb: float = 0.0

    def __post_init__(self):
        if not self.host_threads:
            self.host_threads = hostinfo.default_host_threads(1)
        if not self.host_mem_gb:
            b = hostinfo.host_memory_budget()
            self.host_mem_gb = (b / 2**30) if b else 1e9


@dataclass
class Plan:
    gpu_percentage: int
    prefill_policy: int
    decoding_policy: int
    num_minibatch: int
    n_gpu_layers: int
    prefill_ms: float
    decode_ms_per_step: float
    decode_tokens_per_s: float
    hbm_gb: float
    host_gb: float
    note: str = ""


def layer_bytes(shape):
    return shape.layer_param_bytes()


def estimate(shape, B, T, new, gpu_percentage, decoding_policy, box=None, kv_in_hbm=None):
    """Predicted prefill ms / decode ms per step of one configuration (prefill policy 0 on streamed layers)."""
    box = box or Box()
    L, H, F = shape.layers, shape.hidden, shape.ffn
    n_gpu = int(L * gpu_percentage / 100)
    n_str = L - n_gpu
    lb = layer_bytes(shape)
    S = T + new
    kv_layer = 2 * S * B * H * 2                       # K and V of one layer, bytes
    kv_in_hbm = (decoding_policy == 3) if kv_in_hbm is None else kv_in_hbm
    copy_ms = 1e3 * lb * box.wire_ratio / (box.link_gbs * 1e9)   # bytes shipped per layer in the box's wire format
    # decode
    gemm_ms = 1e3 * lb / (box.hbm_gbs * 1e9)
    attn_gpu_ms = 1e3 * (2 * (T + new // 2) * B * H * 2) / (box.attn_gbs * 1e9)
    attn_host_ms = 1e3 * (2 * (T + new // 2) * B * H * 2) / (min(box.host_threads * box.host_gbs_per_thread, box.host_gbs_cap) * 1e9)
    resident_ms = gemm_ms + attn_gpu_ms
    if decoding_policy == 2:
        streamed_ms = max(copy_ms, gemm_ms + attn_host_ms + 0.3)
    else:
        streamed_ms = max(copy_ms, gemm_ms + attn_gpu_ms)
    lm_ms = 1e3 * shape.vocab * H * 2 / (box.hbm_gbs * 1e9)
    decode_ms = n_gpu * resident_ms + n_str * streamed_ms + lm_ms + 0.5
    # prefill
    flops_layer = 2.0 * B * T * (4 * H * H + 2 * H * F) + 6.0 * B * T * T * H
    pre_layer_ms = 1e3 * flops_layer / (box.mfma_tflops * 1e12)
    prefill_ms = n_gpu * pre_layer_ms + n_str * max(copy_ms, pre_layer_ms) + (pre_layer_ms if n_str else 0.0) + lm_ms
    if n_str and n_gpu:
        prefill_ms += max(0.0, (n_gpu + 1) * pre_layer_ms - 2 * copy_ms)      # head bubble with two slots
    emb_gb = 2 * shape.vocab * H * 2 / 1e9
    hbm = n_gpu * lb / 1e9 + emb_gb + 2 * lb / 1e9 + (L if kv_in_hbm else n_gpu) * kv_layer / 1e9 + 3.5
    host = n_str * lb / 1e9 + (0 if kv_in_hbm else n_str * kv_layer / 1e9)
    return prefill_ms, decode_ms, hbm, host, n_gpu


def plan_cpu_layers(shape, B, T, new, gpu_percentage, box=None, kv_in_hbm=False):
    """How many streamed layers should take their decode step on the host cores (scheduler.forward cpu_layers) beside
    decoding policy 2: a host layer costs its linears at the host's weight-read rate + the host attention, but frees one
    layer's worth of link time.  The step is max(link time of the remaining layers, sum of every layer's latency);
    returns (count, predicted ms per step).  kv_in_hbm: the GPU-computed streamed layers keep their cache in HBM (policy 3 /
    3), so only the host-computed layers use the host cores.  Calibrated on the r01 scans, OPT-30B,
    B = 64, 16 pinned host threads: 16 layers / 412 ms measured with the cache on the host, 19 layers / 367 ms with it in HBM."""
    box = box or Box()
    L, H = shape.layers, shape.hidden
    n_gpu = int(L * gpu_percentage / 100)
    n_str = L - n_gpu
    lb = layer_bytes(shape)
    copy_ms = 1e3 * lb * box.wire_ratio / (box.link_gbs * 1e9)
    gemm_ms = 1e3 * lb / (box.hbm_gbs * 1e9)
    kv_read = 2 * (T + new // 2) * B * H * 2
    attn_gpu_ms = 1e3 * kv_read / (box.attn_gbs * 1e9)
    attn_host_ms = 1e3 * kv_read / (min(box.host_threads * box.host_gbs_per_thread, box.host_gbs_cap) * box.host_attn_beside_stream * 1e9)
    t_g = gemm_ms + (attn_gpu_ms if kv_in_hbm else attn_host_ms) + 0.3
    t_c = 1e3 * lb / (box.host_threads * box.host_linear_gbs_per_thread * 1e9) + attn_host_ms + 0.3
    lm_ms = 1e3 * shape.vocab * H * 2 / (box.hbm_gbs * 1e9)
    best = (0, None)
    for c in range(0, max(1, n_str)):
        step = max((n_str - c) * copy_ms, n_gpu * (gemm_ms + attn_gpu_ms) + (n_str - c) * t_g + c * t_c) + lm_ms + 0.5
        if best[1] is None or step < best[1] - 1e-9:
            best = (c, step)
    return best


def plan(shape, B, T, new, box=None, objective="decode"):
    """Best (gpu%, decode policy) under the HBM and host-memory capacities.  objective: "decode" (tokens/s) or "latency"
    (prefill + new * decode)."""
    box = box or Box()
    best = None
    for pct in range(0, 101, 1):
        if int(shape.layers * pct / 100) == int(shape.layers * (pct - 1) / 100) and pct > 0:
            continue
        for pol in (2, 3):
            pre, dec, hbm, host, n_gpu = estimate(shape, B, T, new, pct, pol, box)
            if hbm > 0.92 * box.hbm_gb or host > 0.85 * box.host_mem_gb:
                continue
            score = dec if objective == "decode" else pre + new * dec
            if best is None or score < best[0] - 1e-9:
                mb = 1
                best = (score, Plan(pct, 3 if pol == 3 else 0, pol, mb, n_gpu, pre, dec, 1e3 * B / dec, hbm, host))
    if best is None:
        raise MemoryError("no placement fits: the model needs more than HBM + host memory allow")
    p = best[1]
    p.note = ("all layers HBM-resident" if p.n_gpu_layers == shape.layers else
              f"{shape.layers - p.n_gpu_layers} layers streamed at {box.link_gbs:.0f} GB/s")
    return p
