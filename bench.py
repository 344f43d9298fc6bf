#!/usr/bin/env python3
"""Benchmark of the LIA hot path on MI355X: OPT-30B, bs=64, in 256 / out 32, gpu%=10,
prefill policy 0 / decode policy 2 (BASELINE.json configs[1], the paper headline).

A "step" is one decode step of the whole batch (every row advances one token) through the offload
scheduler: 4 HBM-resident layers + 44 layers streamed from pinned host memory, GPU linears, host
attention.  The timed region is EXACTLY --steps decode steps after one prefill and --warmup untimed
decode steps, bracketed by barrier + synchronize; value = batch * steps / elapsed (tokens/s).  The
prefill is timed separately (prefill_ms = latency_list[0] of the reference's protocol,
run_generation.py:345-354).  Weights: random-init of the exact architecture (no checkpoints offline).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--model opt-30b] [--batch 64] [--prompt 256]

N > 1 (launched by torch.distributed.run): batch-sharded data parallel, one rank per GPU, each rank owns
`--batch` rows (weak scaling); rank 0 streams every layer once over PCIe and RCCL-broadcasts it over xGMI.
"""
import argparse
import json
import os
import sys
import time

# OpenMP teams park instead of spinning between the per-layer host-attention bursts (a spinning team burns
# the container's CPU quota and the next burst is throttled); must be set before libgomp is mapped.
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
os.environ.setdefault("GOMP_SPINCOUNT", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "isca-2025-lia_amd"))

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_PEAK_TFLOPS = 2500.0  # dense bf16
PCIE_PEAK_GBS = 63.0       # PCIe Gen5 x16 spec


def cpu_baseline(shape, B, T, threads=None):
    """The reference's policy-1 all-CPU path (IPEX/AMX there), timed here as the oracle's CPU restatement
    ("port") on this box's host cores, on a bounded sample: ONE OPT-30B-shaped layer, one decode step at
    S = T+1 and one prefill of B/8 rows, scaled to the full model."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import lia_oracle as orc
    orc.lib()
    from lia_amd import hostinfo
    threads = threads or hostinfo.default_host_threads(1)
    orc.lib().lia_oracle_set_threads(threads)
    fast = bool(orc.lib().lia_oracle_fast_available())
    orc.lib().lia_oracle_set_fast(1 if fast else 0)     # vdpbf16ps inner loops when the host has AVX-512-BF16
    H, F, heads, L = shape.hidden, shape.ffn, shape.heads, shape.layers
    rs = np.random.RandomState(0)
    blk = (rs.standard_normal(1 << 20) * 0.02).astype(np.float32)
    blk_bits = ((blk.view(np.uint32) + 0x8000) >> 16).astype(np.uint16)

    def filled(*shp):
        n = int(np.prod(shp))
        return np.resize(blk_bits, n).reshape(shp)

    W = {"ln1_w": filled(H), "ln1_b": filled(H), "q_w": filled(H, H), "q_b": filled(H), "k_w": filled(H, H),
         "k_b": filled(H), "v_w": filled(H, H), "v_b": filled(H), "out_w": filled(H, H), "out_b": filled(H),
         "ln2_w": filled(H), "ln2_b": filled(H), "fc1_w": filled(F, H), "fc1_b": filled(F), "fc2_w": filled(H, F),
         "fc2_b": filled(H)}
    d = H // heads
    kc, vc = filled(T + 2, B, heads, d).copy(), filled(T + 2, B, heads, d).copy()
    xd = filled(B, 1, H)
    orc.layer_forward(1, W, xd, kc, vc, T, heads)      # warm (page-in, thread pool)
    t0 = time.time()
    reps = 2
    for _ in range(reps):
        orc.layer_forward(1, W, xd, kc, vc, T, heads)
    dec_layer_s = (time.time() - t0) / reps
    Bp = max(1, B // 8)
    xp = filled(Bp, T, H)
    kp, vp = filled(T + 2, Bp, heads, d).copy(), filled(T + 2, Bp, heads, d).copy()
    t0 = time.time()
    orc.layer_forward(1, W, xp, kp, vp, 0, heads)
    pre_layer_s = (time.time() - t0) * (B / Bp)
    # beside it: the product's own policy-1 layer (lia_host_layer_forward, the code `--decoding-policy 1` and --cpu-layers
    # run) on the same sample -- the faster CPU implementation of the two, so the GPU/CPU ratio is not flattered
    product_tps = None
    try:
        import ctypes
        from lia_amd import _native as N, ops
        desc = ops.make_desc(H, heads, F)
        offs, total = ops.pack_offsets(desc)
        flat = np.zeros(total // 2, np.uint16)
        order = ["ln1_w", "ln1_b", "q_w", "q_b", "k_w", "k_b", "v_w", "v_b", "out_w", "out_b", "ln2_w", "ln2_b", "fc1_w", "fc1_b",
                 "fc2_w", "fc2_b"]
        for i, name in enumerate(order):
            a = W[name].reshape(-1)
            flat[offs[i] // 2: offs[i] // 2 + a.size] = a
        wp = ops.weight_ptr_array(flat.ctypes.data, offs)
        yd = np.empty_like(xd)
        args = (ctypes.byref(desc), ctypes.byref(wp), xd.ctypes.data, yd.ctypes.data, kc.ctypes.data, vc.ctypes.data, T + 2, B, B, 1, T, 0,
                threads)
        N.check(N.lib().lia_host_layer_forward(*args))
        t0 = time.time()
        for _ in range(reps):
            N.check(N.lib().lia_host_layer_forward(*args))
        product_tps = B / ((time.time() - t0) / reps * L)
    except Exception as e:          # the baseline of record is the oracle's; this one is informative
        product_tps = f"not measured: {e}"
    return {"value": B / (dec_layer_s * L), "unit": "tokens/s", "cores": threads, "kind": "port",
            "product_host_path_tokens_s": product_tps,
            "cpu": hostinfo.cpu_model(), "isa": hostinfo.isa_flags(), "cpus_usable": hostinfo.usable_cpus(),
            "prefill_ms": 1e3 * pre_layer_s * L,
            "inner_loop": "avx512_bf16 vdpbf16ps" if fast else "fp32 fma",
            "sample": f"oracle policy 1, ONE {shape.name}-shaped layer: decode step B={B} S={T + 1} x{reps} and prefill "
                      f"B={Bp} T={T}; scaled x{L} layers (x{B // Bp} batch for prefill); embeddings/lm_head excluded"}


def pmc_traffic(kernel_substr):
    """HBM bytes per launch of the dominant kernel from the committed PMC pass of this same command
    (profiles/r*_bench_opt30b_pmc_hbm.json: separate --pmc FETCH_SIZE / WRITE_SIZE runs, gfx950 x2 fetch correction).
    PMC collection cannot run inside the timed benchmark, so the live line carries the committed measurement."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_opt30b_pmc_hbm.json")))
    if not files:
        return None, None
    try:
        d = json.load(open(files[-1]))
        # every instantiation of the kernel (128- and 256-row workgroups), weighted by its launches
        hits = [v for k, v in d["kernels"].items() if kernel_substr in k and "traffic_bytes_per_launch" in v]
        calls = sum(v["calls"] for v in hits)
        if calls:
            return sum(v["traffic_bytes_per_launch"] * v["calls"] for v in hits) / calls, os.path.relpath(files[-1], ROOT)
    except (OSError, ValueError, KeyError):
        pass
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=31)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--model", default="opt-30b")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--prompt", type=int, default=256)
    ap.add_argument("--gpu-percentage", type=int, default=10)
    ap.add_argument("--prefill-policy", type=int, default=0)
    ap.add_argument("--decoding-policy", type=int, default=2)
    ap.add_argument("--num-minibatch", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--enable-cxl", action="store_true", help="streamed weights live in the NUMA/CXL tier (numa_alloc_interleave + hipHostRegister)")
    ap.add_argument("--cxl-nodes", default=None, help="NUMA nodes of the CXL tier, e.g. 2,3 (default LIA_CXL_NODES or 2,3)")
    ap.add_argument("--init", default="normal", choices=["normal", "uniform01"])
    ap.add_argument("--stream-format", default=os.environ.get("LIA_STREAM_FORMAT", "pack10"), choices=["raw", "pack12", "pack11", "pack10"],
                    help="wire format of the streamed layers: raw bf16, or the lossless 12-bit / 11.1-bit encodings")
    ap.add_argument("--host-threads", type=int, default=0)
    ap.add_argument("--cpu-layers", type=int, default=0,
                    help="build-defined: with decoding policy 2, this many streamed layers run their decode step on the host cores "
                         "(policy 1 per layer, weights never cross the link); 0 = the reference's uniform policy; -1 = let "
                         "lia_amd.planner.plan_cpu_layers choose from the box's host rates")
    a = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus > 1 and world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} needs torch.distributed.run with {a.gpus} ranks (WORLD_SIZE={world})")
    # LIA_DP_SAME_GPU=1 (validation only, with LIA_DP_BACKEND=gloo): several ranks share one GPU, so the whole
    # batch-shard path (remote tiers, chunked broadcast into staging, decode on non-root ranks) runs on a 1-GPU box
    dev_index = local_rank % max(1, torch.cuda.device_count()) if os.environ.get("LIA_DP_SAME_GPU") == "1" else local_rank
    backend = os.environ.get("LIA_DP_BACKEND", "nccl")
    torch.cuda.set_device(dev_index)
    # the reference pins its CPU work with `numactl -m 0 -C 0-39` (README.md:78); here: the cores of the GPU's NUMA node, where
    # the pinned weights and KV caches live (LIA_PIN_NODE=<n> overrides, -1 = no pinning).  Must precede the first OpenMP team.
    from lia_amd import hostinfo as _hi
    pin_node = int(os.environ["LIA_PIN_NODE"]) if os.environ.get("LIA_PIN_NODE") is not None else _hi.gpu_numa_node(dev_index)
    pinned_cpus = _hi.pin_to_node(pin_node) if pin_node >= 0 else 0
    dist = None
    force_dp = os.environ.get("LIA_FORCE_DP") == "1"      # exercise the broadcast path on a single GPU (world 1)
    if world > 1 or force_dp:
        import torch.distributed as dist
        if "RANK" not in os.environ:
            os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29511")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)

    from lia_amd.model import LiaOPTModel, resolve_shape
    from lia_amd.generation import LIA_KWARGS  # noqa: F401
    from lia_amd.scheduler import KVState, OffloadScheduler
    from lia_amd import dp

    is_llama = "llama" in a.model.lower()
    if is_llama:
        from lia_amd.llama import LiaLlamaModel, LlamaKVState, LlamaScheduler, resolve_llama_shape
        shape = resolve_llama_shape(a.model)
    else:
        shape = resolve_shape(a.model)
    B, T = a.batch, a.prompt
    new = 1 + a.warmup + a.steps
    if T + new > shape.max_pos:
        raise SystemExit("prompt + steps exceeds max positions")
    n_gpu = shape.layers if (is_llama and a.gpu_percentage >= 100) else int(shape.layers * a.gpu_percentage / 100)
    if a.cpu_layers < 0 and not is_llama:
        from lia_amd import planner, hostinfo as _hi2
        a.cpu_layers, _ = planner.plan_cpu_layers(shape, B, T, new, a.gpu_percentage,
                                                  planner.Box(host_threads=a.host_threads or _hi2.default_host_threads(world),
                                                              wire_ratio={"raw": 1.0, "pack12": 0.751, "pack11": 0.696, "pack10": 0.675}[a.stream_format]),
                                                  kv_in_hbm=(a.prefill_policy == 3 and a.decoding_policy == 3))
    flags = dict(prefill_policy=a.prefill_policy, decoding_policy=a.decoding_policy, pin_weight=True,
                 gpu_percentage=a.gpu_percentage, num_minibatch=a.num_minibatch, enable_cxl=a.enable_cxl, no_overlap=False)
    if a.cpu_layers:
        flags["cpu_layers"] = a.cpu_layers
    if a.cxl_nodes:
        from lia_amd.cxl.numa_alloc import set_cxl_nodes
        set_cxl_nodes([int(v) for v in a.cxl_nodes.split(",")])

    t_build = time.time()
    group = dp.DataParallelGroup(dist, rank, world, local_rank) if dist is not None else None
    if group is not None and world > 1:
        group.pin_host_threads()
    if is_llama:
        model = LiaLlamaModel.random_init(shape, seed=0, n_gpu_layers=n_gpu)
        sched = LlamaScheduler(model, device=dev_index)
        KVState = lambda mdl, ng, b, s: LlamaKVState(mdl, b, s)  # noqa: E731,F811
    else:
        pack12 = {"raw": 0, "pack12": 12, "pack11": 11, "pack10": 10}[a.stream_format]
        model = LiaOPTModel.random_init(shape, seed=0, init=a.init, n_gpu_layers=n_gpu, pin_weight=True, enable_cxl=a.enable_cxl,
                                        host_owner=(group is None or group.is_root or group.mode == "allgather"), pack12=pack12,
                                        shard=((rank, world) if (group is not None and world > 1 and group.mode == "allgather") else None),
                                        raw_layers=(OffloadScheduler.cpu_layer_set(n_gpu, shape.layers, a.cpu_layers)
                                                    if (a.cpu_layers and a.decoding_policy in (2, 3) and group is None) else ()))
        sched = OffloadScheduler(model, device=dev_index, dp_group=group, pack12=pack12)
    from lia_amd import hostinfo
    host_threads = a.host_threads or hostinfo.default_host_threads(world)
    g = torch.Generator().manual_seed(0)
    row = torch.randint(4, shape.vocab, (T,), generator=g, dtype=torch.int64)
    row[0] = 2
    ids = row[None, :].repeat(B, 1)                      # identical rows, run_generation.py:285
    build_s = time.time() - t_build

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # untimed shake-out: allocations (pinned KV, workspace), page-in, clocks
    kv = (KVState(model, n_gpu, B, T + new) if is_llama else
          KVState(model, n_gpu, B, T + new, all_on_device=(a.prefill_policy == 3 and a.decoding_policy == 3),
                  host_layers=(OffloadScheduler.cpu_layer_set(n_gpu, shape.layers, a.cpu_layers)
                               if (a.cpu_layers and a.prefill_policy == 3 and a.decoding_policy == 3) else ())))
    sched.forward(ids, kv, max_new_tokens=new, **flags)
    if not is_llama:
        sched.ctx.set_host_threads(host_threads)
        sched.host_threads = host_threads
    cur = ids[:, -1:].clone()
    sched.forward(cur, kv, max_new_tokens=new, **flags)

    # measured generation: prefill, W warm-up decode steps, K timed decode steps
    kv.len = 0
    sched.stream_stats(reset=True)
    sched.ctx.prof_start(4096)
    sync()
    t0 = time.time()
    logits, nxt = sched.forward(ids, kv, max_new_tokens=new, **flags)
    # the reference's first-token latency (greedy_search.py:145,424): wall clock until the iteration's tokens exist.
    # forward() returns after its compute and K/V-delivery streams drained; a device-wide sync here would also wait
    # for the weight prefetch of the NEXT step that the streamer has already started.
    prefill_ms = 1e3 * (time.time() - t0)
    sync()
    prof_prefill = sched.ctx.prof_stop()
    pre_h2d_bytes, pre_h2d_ms = sched.stream_stats()
    cur = nxt.cpu()[:, None]
    for _ in range(a.warmup):
        logits, nxt = sched.forward(cur, kv, max_new_tokens=new, **flags)
        cur = nxt.cpu()[:, None]
    sched.stream_stats(reset=True)
    sched.ctx.prof_start(16384)
    sync()
    thr0 = hostinfo.cgroup_cpu_throttle()
    t0 = time.time()
    step_lat = []
    for _ in range(a.steps):
        ts = time.time()
        logits, nxt = sched.forward(cur, kv, max_new_tokens=new, **flags)
        cur = nxt.cpu()[:, None]
        step_lat.append(time.time() - ts)
    sync()
    elapsed = time.time() - t0
    thr1 = hostinfo.cgroup_cpu_throttle()
    prof = sched.ctx.prof_stop()
    h2d_bytes, h2d_ms = sched.stream_stats()

    if dist is not None:
        tmax = torch.tensor([elapsed, prefill_ms], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed, prefill_ms = float(tmax[0]), float(tmax[1])

    if rank == 0:
        tokens = B * world * a.steps
        sk_n = max(1, prof["skinny_launches"])
        sk_raw_ms = prof["skinny_ms"]
        # the HIP-event bracket reads its own cost too (an empty bracket on the same stream, measured by lia_prof_stop);
        # rocprofv3's kernel durations (profiles/) carry no such term, so it is taken out before dividing
        sk_ms = max(1e-9, sk_raw_ms - sk_n * prof.get("empty_bracket_ms", 0.0))
        achieved = prof["skinny_bytes"] / (sk_ms * 1e-3) / 1e9 if sk_ms > 0 else 0.0
        traffic, traffic_src = (pmc_traffic("lia_gemm_skinny2_kernel<4") if (a.model == "opt-30b" and B == 64) else (None, None))
        out = {
            "metric": "decode tokens/s (+ prefill ms), OPT-30B bs=64 in256/out32 gpu%=10" if (a.model == "opt-30b" and B == 64 and T == 256 and a.gpu_percentage == 10 and not a.cpu_layers)
                      else f"decode tokens/s (+ prefill ms), {a.model} bs={B} in{T} gpu%={a.gpu_percentage}",
            "value": tokens / elapsed, "unit": "tokens/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * elapsed / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"{shape.name} shape (random-init N(0,0.02)), batch {B}/GPU identical rows, prompt {T}, "
                                   f"gpu%={a.gpu_percentage} ({n_gpu} resident + {shape.layers - n_gpu} streamed layers), "
                                   f"prefill policy {a.prefill_policy}, decode policy {a.decoding_policy}, pin-weight{', enable-cxl nodes ' + str(a.cxl_nodes) if a.enable_cxl else ''}, "
                                   f"num-minibatch {a.num_minibatch}{', ' + str(a.cpu_layers) + ' decode layers on the host cores' if a.cpu_layers else ''}",
                       "global_batch": B * world, "prompt_len": T, "new_tokens": new,
                       "parallelism": (f"dp{world} batch-shard, {group.mode} weight stream" if world > 1 else "single GPU"),
                       "host_attention_threads": host_threads, "host_numa_node": pin_node if pinned_cpus else None},
            "prefill_ms": prefill_ms,
            "decode_latency_ms": {"mean": 1e3 * sum(step_lat) / len(step_lat), "p90": 1e3 * sorted(step_lat)[int(0.9 * (len(step_lat) - 1))],
                                  "max": 1e3 * max(step_lat)},
            "roofline": {"bound": "hbm", "kernel": "lia_gemm_skinny2_kernel<4,3,1,8,RT> (RT = 1 and 2; decode linears + lm_head)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src, "launches": prof["skinny_launches"], "avg_launch_us": 1e3 * sk_ms / sk_n,
                         "avg_bracket_us_raw": 1e3 * sk_raw_ms / sk_n, "empty_bracket_us": 1e3 * prof.get("empty_bracket_ms", 0.0),
                         "algorithmic_bytes_per_launch": prof["skinny_bytes"] / sk_n},
            "host_link": {"bound": "pcie", "stream_format": a.stream_format if not is_llama else "raw",
                          "weight_bytes_per_step": float(getattr(model, "streamed_bytes", lambda n: 0)(n_gpu)) if not is_llama else None,
                          "achieved": h2d_bytes / (elapsed * 1e9), "peak": PCIE_PEAK_GBS, "unit": "GB/s",
                          "frac": h2d_bytes / (elapsed * 1e9) / PCIE_PEAK_GBS,
                          "copy_engine_busy_frac": (h2d_ms * 1e-3) / elapsed, "bytes_per_step": h2d_bytes / a.steps},
            "prefill_detail": {"gemm_ms": prof_prefill["tiled_ms"], "gemm_launches": prof_prefill["tiled_launches"],
                               "gemm_tflops": prof_prefill["tiled_flops"] / max(prof_prefill["tiled_ms"], 1e-9) / 1e9,
                               "mfma_frac": prof_prefill["tiled_flops"] / max(prof_prefill["tiled_ms"], 1e-9) / 1e9 / MFMA_PEAK_TFLOPS,
                               "h2d_busy_ms": pre_h2d_ms, "h2d_gbs_while_busy": pre_h2d_bytes / max(pre_h2d_ms, 1e-9) / 1e6},
            "host_cpu_throttle": {"periods": thr1[0] - thr0[0], "throttled_ms": (thr1[1] - thr0[1]) / 1e3,
                                  "note": "cgroup CFS quota stalls during the timed decode steps (cpu.stat)"},
            "build_s": build_s,
            "host_memory_gib": {k: (None if v is None else round(v / 2**30, 2)) for k, v in hostinfo.cgroup_memory().items()},
        }
        if world == 1 and not a.no_cpu_baseline and not is_llama:
            out["cpu_baseline"] = cpu_baseline(shape, B, T)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # RCCL writes its version banner through C stdio, which is block-buffered when stdout is a pipe: flush it
        # first so that the JSON line is the LAST line of the output
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
