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
    host_linear_gbs_per_thread: float = 7.0   # policy-1 linears at M = 64 beside the running weight stream (7.3 alone; r02's kernel: 4.4 / 4.9; 3.6 unpinned)
    host_attn_beside_stream: float = 0.8      # share of its rate the host attention keeps beside the stream (0.55 unpinned)
    wire_ratio: float = 0.675         # bytes shipped per weight byte: pack10 0.675, raw 1.0
    host_threads: int = 0
    host_mem_gb: float = 0.0
    calibrated: dict = None           # set by calibrate(): the measured numbers, for the log

    def __post_init__(self):
        if not self.host_threads:
            self.host_threads = hostinfo.default_host_threads(1)
        if not self.host_mem_gb:
            b = hostinfo.host_memory_budget()
            self.host_mem_gb = (b / 2**30) if b else 1e9


def calibrate(device=0, host_threads=0, budget_s=8.0, verbose=False):
    """Measure this box instead of trusting the defaults above (the reference hand-picks per script
    line, llm/scripts/lia_offline.sh:13-29): pinned H2D rate, the decode GEMM's weight-read rate at M = 64, the decode
    attention's KV-read rate, the prefill GEMM's TFLOP/s, and the host attention / host linear rates at the thread count the
    scheduler will use -- every number through the same C ABI entry points the hot path calls.  Takes a few seconds."""
    import ctypes
    import time

    import numpy as np
    import torch

    from . import _native as N, ops
    L = N.lib()
    threads = host_threads or hostinfo.default_host_threads(1)
    box = Box(host_threads=threads)
    t_start = time.time()
    torch.cuda.set_device(device)
    ctx = ops.Context(device, 1 << 30)

    def timed(fn, reps):
        fn()
        ctx.synchronize()
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(reps):
            fn()
        ctx.synchronize()
        torch.cuda.synchronize()
        return (time.time() - t0) / reps

    g = torch.Generator(device="cuda").manual_seed(1)
    rnd = lambda *s: (0.02 * torch.randn(*s, generator=g, device="cuda")).to(torch.bfloat16)  # noqa: E731
    # host link: one pinned 256 MB block, blocking copies
    nb = 256 << 20
    hp = L.lia_host_alloc_pinned(nb)
    if not hp:
        raise MemoryError("calibrate: cannot pin 256 MB")
    dbuf = torch.empty(nb, dtype=torch.uint8, device="cuda")
    dt = timed(lambda: N.check(L.lia_memcpy_h2d(ctypes.c_void_p(dbuf.data_ptr()), ctypes.c_void_p(hp), nb)), 3)
    box.link_gbs = nb / dt / 1e9
    L.lia_host_free_pinned(hp)
    del dbuf
    # decode GEMM at M = 64: four weight buffers of 117 MB take turns (beyond the 256 MB Infinity Cache)
    K, Nw = 7168, 8192
    ws = [rnd(Nw, K) for _ in range(4)]
    x = rnd(64, K)
    it = [0]

    def skinny():
        ctx.linear(x, ws[it[0] % 4])
        it[0] += 1
    dt = timed(skinny, 12)
    box.hbm_gbs = 2.0 * Nw * K / dt / 1e9
    # prefill GEMM
    xb = rnd(4096, K)
    dt = timed(lambda: ctx.linear(xb, ws[0]), 4)
    box.mfma_tflops = 2.0 * 4096 * Nw * K / dt / 1e12
    del ws, xb
    # decode attention over an HBM-resident cache
    B, S, heads, d = 64, 512, 56, 128
    kc = rnd(S, B, heads, d)
    vc = rnd(S, B, heads, d)
    q = rnd(B, 1, heads * d)
    dt = timed(lambda: ctx.attention(q, kc, vc, S, heads), 6)
    box.attn_gbs = 2.0 * S * B * heads * d * 2 / dt / 1e9
    del kc, vc, q
    # host attention and host linear at the scheduler's thread count (numpy buffers: pageable host memory is what they read)
    Bh, Sh = 16, 256
    hk = np.full((Sh + 1, Bh, heads, d), 0x3c00, np.uint16)      # (np.zeros maps every page to the one zero page: an L1-resident cache)
    hv = hk.copy()
    hq = np.zeros((Bh, 1, heads * d), np.uint16)
    ho = np.zeros_like(hq)
    args = (hq.ctypes.data, hq.ctypes.data, hq.ctypes.data, hk.ctypes.data, hv.ctypes.data, ho.ctypes.data, Bh, 1, Sh, heads, d, Bh, 0, threads)
    N.check(L.lia_host_attention(*args))
    t0 = time.time()
    reps = 6
    for _ in range(reps):
        N.check(L.lia_host_attention(*args))
    rate = 2.0 * Sh * Bh * heads * d * 2 / ((time.time() - t0) / reps) / 1e9
    box.host_gbs_per_thread, box.host_gbs_cap = rate / threads, rate
    if L.lia_host_has_avx512_bf16() and time.time() - t_start < budget_s:
        Nh = 16384                      # 235 MB of weights at K = 7168: beyond the L3 slices of the threads' CCDs, like a real layer
        hw = np.full((Nh, K), 0x3c00, np.uint16)
        hx = np.zeros((64, K), np.uint16)
        hy = np.zeros((64, Nh), np.uint16)
        largs = (hx.ctypes.data, hw.ctypes.data, None, None, hy.ctypes.data, 64, Nh, K, 0, threads)
        N.check(L.lia_host_linear(*largs))
        t0 = time.time()
        for _ in range(3):
            N.check(L.lia_host_linear(*largs))
        # (x 0.95: what the linears keep beside the running weight stream -- 7.3 alone vs 7.0 fitted on the r03 scans)
        box.host_linear_gbs_per_thread = 0.95 * 2.0 * Nh * K / ((time.time() - t0) / 3) / 1e9 / threads
    ctx.close()
    box.calibrated = {"seconds": round(time.time() - t_start, 2), "link_gbs": round(box.link_gbs, 1), "hbm_gbs": round(box.hbm_gbs),
                      "attn_gbs": round(box.attn_gbs), "mfma_tflops": round(box.mfma_tflops), "host_attention_gbs": round(rate, 1),
                      "host_linear_gbs_per_thread": round(box.host_linear_gbs_per_thread, 2), "host_threads": threads}
    if verbose:
        print("planner.calibrate:", box.calibrated)
    return box


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


def decode_gemm_ms(shape, B, box):
    """the four linears of one decode step of one layer: weight-read-bound up to M = 256 (the skinny kernels), MFMA-bound beyond
    (r06: the tiled kernel at 256 < M < 1024 does ~0.65 of the large-M rate, tools/gemm_bench -- M = 900: 0.9 of 1.45 PFLOP/s)"""
    lb = layer_bytes(shape)
    bw_ms = 1e3 * lb / (box.hbm_gbs * 1e9)
    if B <= 256:
        return bw_ms
    return max(bw_ms, 1e3 * 2.0 * B * (lb / 2) / (0.65 * box.mfma_tflops * 1e12))


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
    gemm_ms = decode_gemm_ms(shape, B, box)
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
    # HBM beside the resident layers and the caches: four streamer slots + their wire-format staging areas, the context's workspace
    # for the prefill's B x T rows (lia_api.hip ws_layout: 6 row buffers of H, one of F, two K/V slabs of 2 H), the two hidden-state
    # buffers, lm_head's logits.  (r06: the estimate carried two slots and a constant; at --batch-size 900 the planner's pick ran
    # out of HBM in lia_ctx_create -- results/r06_matrix_offline_opt30b_32_256_b900_p02_g0_autoplan.json of the first attempt)
    rows = B * T
    ws_gb = (rows * (10 * H + F) * 2 + 8 * min(rows, 256) * max(3 * H, F) * 4) / 1e9
    stream_gb = 4 * lb * (1.0 + (1.13 if box.wire_ratio < 1.0 else 0.0)) / 1e9 if n_str else 0.0      # (staging = lia_pack10_bound: 1.13 x the raw layer)
    hbm = (n_gpu * lb / 1e9 + emb_gb + stream_gb + (L if kv_in_hbm else n_gpu) * kv_layer / 1e9 + ws_gb + 2 * rows * H * 2 / 1e9 +
           B * shape.vocab * 6 / 1e9 + 3.5)
    host = n_str * lb / 1e9 + (0 if kv_in_hbm else n_str * kv_layer / 1e9)
    return prefill_ms, decode_ms, hbm, host, n_gpu


def plan_cpu_layers(shape, B, T, new, gpu_percentage, box=None, kv_in_hbm=False):
    """How many streamed layers should take their decode step on the host cores (scheduler.forward cpu_layers) beside
    decoding policy 2: a host layer costs its linears at the host's weight-read rate + the host attention, but frees one
    layer's worth of link time.  The step is max(link time of the remaining layers, sum of every layer's latency);
    returns (count, predicted ms per step).  kv_in_hbm: the GPU-computed streamed layers keep their cache in HBM (policy 3 /
    3), so only the host-computed layers use the host cores.  Calibrated on the r03 scans, OPT-30B,
    B = 64, 16 pinned host threads: 18-19 layers / 384-390 ms measured with the cache on the host, 21-23 layers / 318-335 ms with it in
    HBM (r01-r02, before the second pass over the host linears: 16 / 412 and 19 / 367)."""
    box = box or Box()
    L, H = shape.layers, shape.hidden
    n_gpu = int(L * gpu_percentage / 100)
    n_str = L - n_gpu
    lb = layer_bytes(shape)
    copy_ms = 1e3 * lb * box.wire_ratio / (box.link_gbs * 1e9)
    gemm_ms = decode_gemm_ms(shape, B, box)
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


def plan(shape, B, T, new, box=None, objective="decode", max_gpu_percentage=100):
    """Best (gpu%, decode policy) under the HBM and host-memory capacities.  objective: "decode" (tokens/s) or "latency"
    (prefill + new * decode).  max_gpu_percentage: an upper bound on the resident share (what-if runs, tests)."""
    box = box or Box()
    best = None
    for pct in range(0, min(100, int(max_gpu_percentage)) + 1, 1):
        if int(shape.layers * pct / 100) == int(shape.layers * (pct - 1) / 100) and pct > 0:
            continue
        for pol in (2, 3):
            pre, dec, hbm, host, n_gpu = estimate(shape, B, T, new, pct, pol, box)
            if hbm > 0.90 * box.hbm_gb or host > 0.85 * box.host_mem_gb:
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


# ---------------------------------------------------------------------------------------------------------------------
# N > 1 (BASELINE.json configs[4]): a PREDICTION of the batch-sharded decode step, written down before
# any multi-GPU hardware has run this code, so that the first SCALE_r*.json can be read against it.
# ---------------------------------------------------------------------------------------------------------------------
XGMI_LINK_GBS = 153.0          # one xGMI link, per direction (MI355X: 7 links per GPU, point to point)
RCCL_RING_EFF = 0.70           # share of a link a pipelined ring broadcast / all-gather sustains on 16-64 MB chunks (assumed; never measured here)
HOST_DRAM_GBS = 400.0          # what one socket's DRAM feeds to N copy engines at once (assumed; the r03 box read 223 GB/s with 16 threads)
WIRE_DECODE_MS_PER_GB = 0.47   # lia_pack10_decode_kernel: 392 us for an OPT-30B layer's 0.832 GB of wire bytes (profiles/r05_opt30b_decode_timeline.txt)


def predict_dp(shape, rows_per_rank, T, new, gpu_percentage, world, mode="broadcast", decoding_policy=3, box=None, host_cpus=16):
    """Predicted decode step of `world` ranks on one node, each with `rows_per_rank` rows: (ms per step, the binding resource,
    every term).  mode: "broadcast" (the north star's: the root's link carries every streamed layer once, RCCL broadcasts it) or
    "allgather" (every rank pulls 1 / world of the wire bytes over ITS link, one all-gather per layer).  decoding_policy 3: every
    layer's KV cache in the rank's HBM (bench.py's N > 1 default); 2: host attention on the rank's share of the host cores.
    The terms: the link (or links), the collective on one xGMI link, the wire decode + GEMMs + attention on each GPU, the host
    attention on host_cpus / world threads, and the host memory the pinned weights (+ caches) need."""
    box = box or Box()
    L, H = shape.layers, shape.hidden
    n_gpu = int(L * gpu_percentage / 100)
    n_str = L - n_gpu
    lb = layer_bytes(shape)
    wire = lb * box.wire_ratio
    kv_read = 2 * (T + new // 2) * rows_per_rank * H * 2
    gemm_ms = decode_gemm_ms(shape, rows_per_rank, box)
    attn_gpu_ms = 1e3 * kv_read / (box.attn_gbs * 1e9)
    threads = max(1, host_cpus // world)
    attn_host_ms = 1e3 * kv_read / (min(threads * box.host_gbs_per_thread, box.host_gbs_cap) * box.host_attn_beside_stream * 1e9)
    decode_ms = WIRE_DECODE_MS_PER_GB * wire / 1e9 if box.wire_ratio < 1.0 else 0.0
    lm_ms = 1e3 * shape.vocab * H * 2 / (box.hbm_gbs * 1e9)
    if mode == "allgather" and world > 1:
        link_ms = n_str * 1e3 * wire / (min(world * box.link_gbs, HOST_DRAM_GBS) * 1e9)
        coll_ms = n_str * 1e3 * wire * (world - 1) / world / (XGMI_LINK_GBS * RCCL_RING_EFF * 1e9)
    else:
        link_ms = n_str * 1e3 * wire / (box.link_gbs * 1e9)
        coll_ms = n_str * 1e3 * wire / (XGMI_LINK_GBS * RCCL_RING_EFF * 1e9) if world > 1 else 0.0
    # RCCL on a fully connected xGMI node may spread a collective over up to min(world - 1, 7) links instead of one ring: the
    # optimistic end of the prediction (never measured here either)
    coll_all_links_ms = coll_ms / max(1, min(world - 1, 7))
    per_layer_gpu = gemm_ms + decode_ms + (attn_gpu_ms if decoding_policy == 3 else attn_host_ms + 0.3)
    gpu_ms = n_gpu * (gemm_ms + attn_gpu_ms) + n_str * per_layer_gpu + lm_ms + 0.5
    terms = {"host_link_ms": link_ms, "xgmi_collective_ms": coll_ms, "xgmi_collective_all_links_ms": coll_all_links_ms,
             "ms_per_step_if_all_links": max(link_ms, coll_all_links_ms, gpu_ms), "per_gpu_compute_ms": gpu_ms,
             "host_attention_ms": (n_str * attn_host_ms if decoding_policy == 2 else 0.0), "host_attention_threads_per_rank": threads}
    # the copy stream runs the host copy and the collective back to back per chunk (scheduler._prefetch_broadcast): they pipeline
    # across chunks, so the stream's time is the larger of the two, and the step is the larger of that and the compute chain
    step = max(link_ms, coll_ms, gpu_ms)
    bound = max((("host link (root's PCIe)" if mode != "allgather" else "host links (one per rank) / host DRAM"), link_ms),
                ("RCCL collective on one xGMI link", coll_ms),
                (("host attention threads" if decoding_policy == 2 and n_str * attn_host_ms > 0.5 * gpu_ms else "per-GPU compute"), gpu_ms),
                key=lambda t: t[1])[0]
    kv_gb = 2 * (T + new) * rows_per_rank * H * 2 * L / 1e9
    terms["pinned_host_gb"] = n_str * wire / 1e9 + (world * kv_gb * n_str / L if decoding_policy == 2 else 0.0)
    terms["hbm_gb_per_rank"] = n_gpu * lb / 1e9 + 4 * lb / 1e9 + (kv_gb if decoding_policy == 3 else kv_gb * n_gpu / L) + 2 * shape.vocab * H * 2 / 1e9 + 3.5
    return step, bound, terms


def predict_dp_table(shape, T, new, gpu_percentage, box=None, rows_weak=64, global_batch=256, host_cpus=16):
    """the table DESIGN.md section 6 and bench.py's `config.predicted` carry: N = 1, 2, 4, 8 x {weak: rows_weak rows per GPU, strong:
    global_batch rows over the GPUs} x {broadcast, allgather} x {policy 3 / 3, policy 0 / 2}"""
    rows = []
    for scaling, rows_of in (("weak", lambda n: rows_weak), ("strong", lambda n: -(-global_batch // n))):
        for n in (1, 2, 4, 8):
            for mode in ("broadcast", "allgather"):
                if n == 1 and mode == "allgather":
                    continue
                for pol in (3, 2):
                    r = rows_of(n)
                    ms, bound, terms = predict_dp(shape, r, T, new, gpu_percentage, n, mode, pol, box, host_cpus)
                    rows.append({"scaling": scaling, "n_gpus": n, "rows_per_gpu": r, "mode": mode, "policies": "3/3" if pol == 3 else "0/2",
                                 "ms_per_step": round(ms, 1), "tokens_per_s": round(1e3 * r * n / ms, 1), "bound_by": bound,
                                 **{k: (round(v, 1) if isinstance(v, float) else v) for k, v in terms.items()}})
    return rows
