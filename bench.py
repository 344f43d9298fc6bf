#!/usr/bin/env python3
"""Benchmark of the LIA hot path on MI355X: OPT-30B, bs=64, in 256 / out 32, gpu%=10,
prefill policy 0 / decode policy 2 (BASELINE.json configs[1], the paper headline).

The measurement IS the reference's protocol (llm/single_instance/run_generation.py:308-354): one greedy
`generate(..., token_latency=True)` over the harness's identical-row batch, `max_new_tokens = 1 + warmup + steps`
(default 1 + 0 + 31 = the config's 32 new tokens), `latency_list[0]` = prefill, `latency_list[1:]` = decode steps.
A "step" is one decode step of the whole batch through the offload scheduler (4 HBM-resident layers + 44 layers
streamed from pinned host memory, GPU linears, host attention).  Around EXACTLY the last --steps decode steps a
`step_hook` puts barrier + device synchronize and the HIP-event brackets of the roofline object;
value = batch x steps / that bracket (tokens/s), protocol.decode_tokens_per_s = batch / mean(latency_list[1:]).
An untimed warm-up generate (same cache size, two decode steps) runs first, as --num-warmup iterations do there.
Weights: seeded random init of the exact architecture (no checkpoints offline).

In the same run, on rank 0 at N = 1: a shorter leg with the streamed layers in RAW bf16 (what the reference
ships; value_raw_format) and the CPU baseline -- decode policy 1 (every layer on the host cores) through the same
generate() entry point, beside the oracle's restatement of that path on a one-layer sample.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--model opt-30b] [--batch 64 | --global-batch 256] [--prompt 256]

N > 1: batch-sharded data parallel, one rank per GPU over RCCL; started as-is it launches its own
`python -m torch.distributed.run --nproc-per-node N` children (before touching any GPU) and relays rank 0's line.
"""
import argparse
import json
import math
import os
import socket
import subprocess
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
WIRE = {"raw": 0, "pack10": 10}


def build_parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=31)
    ap.add_argument("--warmup", type=int, default=0, help="untimed decode steps of the timed generate() before the bracket "
                    "(a separate warm-up generate always runs first)")
    ap.add_argument("--model", default="opt-30b")
    ap.add_argument("--batch", type=int, default=64, help="rows per GPU (weak scaling)")
    ap.add_argument("--global-batch", type=int, default=0, help="total rows, split evenly over the GPUs (BASELINE config 5: 256 over 8)")
    ap.add_argument("--prompt", type=int, default=256)
    ap.add_argument("--gpu-percentage", type=int, default=10)
    ap.add_argument("--prefill-policy", type=int, default=None, help="default: 0 on one GPU (BASELINE configs[1]); on N > 1 GPUs 3 when every "
                    "rank's KV cache fits its HBM (plan_policies), else 0")
    ap.add_argument("--decoding-policy", type=int, default=None, help="default: 2 on one GPU; on N > 1 GPUs 3 (KV in HBM: G ranks share the host's "
                    "CPU quota, host attention would run on 1/G of it each), else 2")
    ap.add_argument("--num-minibatch", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-raw-leg", action="store_true", help="skip the second, shorter leg with raw bf16 on the wire")
    ap.add_argument("--raw-steps", type=int, default=6)
    ap.add_argument("--no-defer-kv-leg", action="store_true", help="skip the two extra prefills with LIA_DEFER_KV=0 (prefill_ms_defer_kv_0)")
    ap.add_argument("--no-cooperative-leg", action="store_true", help="skip the build-defined cooperative-split leg (value_cooperative)")
    ap.add_argument("--no-auto-plan", action="store_true", help="skip the auto_plan object (what run.py --auto-plan would choose on this box, ~5 s)")
    ap.add_argument("--no-cooperative-kv-leg", action="store_true", help="skip the cooperative split's KV-in-HBM variant (value_cooperative_kv_in_hbm)")
    ap.add_argument("--coop-steps", type=int, default=28, help="decode steps the cooperative leg gives the controller's search (it takes 12-20); "
                    "--coop-windows x 8 more steps follow, value_cooperative = the median window")
    ap.add_argument("--coop-windows", type=int, default=3, help="8-step windows behind the search; value_cooperative* = their median rate, each window "
                    "reported with its own CFS-throttle counters (r05 verdict, weak item 9: one 8-step average had an unknown spread)")
    ap.add_argument("--no-dp-extra-legs", action="store_true", help="N > 1: skip the KV-in-HBM and all-gather legs")
    ap.add_argument("--dp-allgather-legs", action="store_true",
                    help="N > 1: also run the all-gather legs (every rank re-draws the model and pins 1/N of each streamed layer; value_allgather*). "
                         "Off by default: this streaming mode has run over gloo and over RCCL at world size 1 only, and an optional leg must "
                         "not be able to take the measured headline line down with it on first contact with an 8-GPU node")
    ap.add_argument("--dp-extra-steps", type=int, default=6)
    ap.add_argument("--dp-backend", default="nccl", choices=["nccl", "gloo"], help="collective backend (nccl = RCCL; gloo for dry runs with --dp-same-gpu)")
    ap.add_argument("--dp-same-gpu", action="store_true", help="dry run: the ranks share GPU 0 (one-GPU box, --dp-backend gloo)")
    ap.add_argument("--force-dp", action="store_true", help="dry run: take the data-parallel (broadcast) path at world size 1")
    ap.add_argument("--dp-extra-timeout", type=int, default=420, help="seconds the extra legs may take before the run ends with the headline line only")
    ap.add_argument("--cpu-steps", type=int, default=4, help="decode steps of the policy-1 CPU baseline leg")
    ap.add_argument("--cpu-prefill-layers", type=int, default=2, help="layers of the policy-1 CPU prefill measured through lia_host_layer_forward "
                    "at the configuration's B x T (cpu_baseline.prefill; 0 = keep the oracle's scaled one-layer sample only)")
    ap.add_argument("--enable-cxl", action="store_true", help="streamed weights live in the NUMA/CXL tier (numa_alloc_interleave + hipHostRegister)")
    ap.add_argument("--cxl-nodes", default=None, help="NUMA nodes of the CXL tier, e.g. 2,3 (default LIA_CXL_NODES or 2,3)")
    ap.add_argument("--init", default="normal", choices=["normal", "uniform01", "trained-like"],
                    help="normal = HF _init_weights N(0, 0.02) (the headline); uniform01 = the reference's dummy-weight recipe; trained-like = "
                         "per-tensor scales over ~3 binades, 0.1 %% outlier channels at 20 sigma, LayerNorm gains near 1 (a stress of the wire "
                         "format: config.bits_per_value and layers_shipped_raw say what it cost)")
    ap.add_argument("--stream-format", default=None, choices=sorted(WIRE),
                    help="wire format of the streamed layers: raw bf16, or a lossless packed encoding (default: lia_amd.scheduler."
                         "default_stream_format() = $LIA_STREAM_FORMAT or pack10 -- the same default as run.py / OffloadScheduler)")
    ap.add_argument("--host-threads", type=int, default=0)
    ap.add_argument("--bracket-stride", type=int, default=0, help="0 = 8 when layers stream (the step is link-bound, a bracket's gaps are free), "
                    "32 when every layer is resident (r04: the 16-24 brackets per step at stride 8 cost 0.2-0.3 ms of a 7.7 / 16 ms step); "
                    "HIP-event bracket around every Nth decode GEMM launch (a bracket "
                    "costs two ~6 us idle gaps on the stream; 8 is co-prime to the 193 / 129 launches of an OPT-30B / Llama-3-8B step)")
    ap.add_argument("--cpu-layers", type=int, default=0,
                    help="build-defined: with decoding policy 2, this many streamed layers run their decode step on the host cores "
                         "(policy 1 per layer, weights never cross the link); 0 = the reference's uniform policy; -1 = let "
                         "lia_amd.planner.plan_cpu_layers choose from the box's host rates")
    ap.add_argument("--selftest-launcher", action="store_true", help=argparse.SUPPRESS)   # CPU test of the --gpus N self-launch
    ap.add_argument("--selftest-dp-line", action="store_true", help=argparse.SUPPRESS)    # CPU (gloo) test of the N > 1 parts of the line
    return ap


# ---------------------------------------------------------------------------------------------------------------------
# --gpus N without a launcher: start the ranks ourselves.  This process has not imported torch or touched a GPU yet,
# and it never replaces itself: the ranks are children, their output is relayed, their exit code is ours.
# ---------------------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(n, argv):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: what RCCL needs on this pool
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, bufsize=1)
    last_json = None
    for line in proc.stdout:
        s = line.rstrip("\n")
        if s.startswith("{") and s.endswith("}") and ('"metric"' in s or '"launcher_selftest"' in s):
            last_json = s
        else:
            print(s, flush=True)
    rc = proc.wait()
    if last_json is not None:
        print(last_json, flush=True)        # rank 0's line is the LAST line of our output
    if rc != 0:
        print(f"bench.py: a rank exited with code {rc}", file=sys.stderr)
    return rc if rc != 0 else (0 if last_json is not None else 1)


def launcher_selftest():
    """CPU-only (gloo) body for the self-launch test: every rank joins, one all-reduce, rank 0 prints one JSON line."""
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo")
    t = torch.tensor([dist.get_rank() + 1.0])
    dist.all_reduce(t)
    world = dist.get_world_size()
    dist.barrier()
    if dist.get_rank() == 0:
        print(json.dumps({"launcher_selftest": True, "world": world, "sum": float(t[0])}), flush=True)
    dist.destroy_process_group()
    return 0


# ---------------------------------------------------------------------------------------------------------------------
def cpu_oracle_sample(shape, B, T, threads):
    """The oracle's CPU restatement of the policy-1 path ("port") on a bounded sample: ONE layer of the model's shape, decode
    step at S = T+1 (x2) and a prefill of B/8 rows, scaled to the full model."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import lia_oracle as orc
    orc.lib()
    orc.lib().lia_oracle_set_threads(threads)
    fast = bool(orc.lib().lia_oracle_fast_available())
    orc.lib().lia_oracle_set_fast(1 if fast else 0)     # vdpbf16ps inner loops when the host has AVX-512-BF16
    H, F, heads, L = shape.hidden, shape.ffn, shape.heads, shape.layers
    rs = np.random.RandomState(0)
    blk = (rs.standard_normal(1 << 20) * 0.02).astype(np.float32)
    blk_bits = ((blk.view(np.uint32) + 0x8000) >> 16).astype(np.uint16)

    def filled(*shp):
        return np.resize(blk_bits, int(np.prod(shp))).reshape(shp)

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
    return {"decode_tokens_per_s": B / (dec_layer_s * L), "prefill_ms": 1e3 * pre_layer_s * L,
            "inner_loop": "avx512_bf16 vdpbf16ps" if fast else "fp32 fma",
            "sample": f"oracle policy 1, ONE {shape.name}-shaped layer: decode step B={B} S={T + 1} x{reps} and prefill B={Bp} T={T}; "
                      f"scaled x{L} layers (x{B // Bp} batch for prefill); embeddings / lm_head excluded"}


def cpu_host_head_sample(model, shape, B, threads, reps=3):
    """The part of a decode step the policy-1 leg leaves on the GPU -- token + position embedding, final LayerNorm, the tied lm_head on
    the last position, argmax -- measured on the HOST cores (the reference's policy 1 runs all of it on the CPU: embeddings
    lia/modeling_opt.py:1108,1137-1142, final LN :1563, lm_head models.py:424-431): lia_host_layernorm + lia_host_linear over a host
    copy of the embedding matrix (0.72 GB for OPT-30B) + numpy gather / argmax.  Returns ms per decode step; cpu_baseline.value adds
    it to every host-layer step, so the reported baseline is a FULL-CPU step."""
    import ctypes
    import numpy as np
    import torch
    from lia_amd import _native as N
    L = N.lib()
    H, V = shape.hidden, shape.vocab
    tok = model.embed_tokens.cpu().contiguous().view(torch.int16).numpy().view(np.uint16)
    pos = model.embed_positions.cpu().contiguous().view(torch.int16).numpy().view(np.uint16)
    lnw = model.final_ln_w.cpu().contiguous().view(torch.int16).numpy().view(np.uint16)
    lnb = model.final_ln_b.cpu().contiguous().view(torch.int16).numpy().view(np.uint16)
    ids = np.random.RandomState(1).randint(4, V, size=B)
    f32 = lambda b: (b.astype(np.uint32) << 16).view(np.float32)          # noqa: E731
    bf = lambda f: ((f.view(np.uint32) + 0x8000) >> 16).astype(np.uint16)  # noqa: E731
    hid, y, logits = np.empty((B, H), np.uint16), np.empty((B, H), np.uint16), np.empty((B, V), np.uint16)
    times = []
    for _ in range(reps + 1):
        t0 = time.time()
        hid[:] = bf(f32(tok[ids]) + f32(pos[258][None, :]))                 # embed_tokens[ids] + embed_positions[position + 2]
        N.check(L.lia_host_layernorm(hid.ctypes.data, lnw.ctypes.data, lnb.ctypes.data, y.ctypes.data, B, H, ctypes.c_float(shape.ln_eps), threads))
        N.check(L.lia_host_linear(y.ctypes.data, tok.ctypes.data, None, None, logits.ctypes.data, B, V, H, 0, threads))
        nxt = f32(logits).argmax(-1)
        times.append(time.time() - t0)
    return {"ms_per_step": 1e3 * min(times[1:]), "runs_ms": [round(1e3 * t, 2) for t in times[1:]], "threads": threads, "argmax_checksum": int(nxt.sum()),
            "what": f"embedding gather + final LayerNorm + lm_head [{B} x {H}] x [{V} x {H}]^T + argmax on the host cores (lia_host_layernorm, lia_host_linear)"}


def cpu_product_prefill_sample(model, shape, B, T, threads, n_layers=2):
    """The CPU baseline's PREFILL through the product's own host path (r05, r04 verdict item 5): lia_host_layer_forward on the
    configuration's full B x T rows (M = B * T > 256 -> the generic host GEMM, not the decode kernel) for n_layers consecutive
    "layers" (one drawn layer's weights, each call fed the previous call's output and its own KV cache), scaled to the model's
    layer count.  ~20 TFLOP per OPT-30B layer: a few seconds per layer on 16 cores."""
    import ctypes
    import numpy as np
    import torch
    from lia_amd import _native as N, ops
    from lia_amd.model import draw_layer
    L = N.lib()
    H, F, heads = shape.hidden, shape.ffn, shape.heads
    d = H // heads
    flat = draw_layer(shape, model.offsets, model.layer_bytes, li=5, seed=321).cpu()
    torch.cuda.synchronize()
    host = flat.view(torch.int16).numpy().view(np.uint16)
    arr = (ctypes.c_void_p * 16)(*[host.ctypes.data + off for off in model.offsets])
    rs = np.random.RandomState(3)
    blk = (rs.standard_normal(1 << 20)).astype(np.float32)
    x = np.resize(((blk.view(np.uint32) + 0x8000) >> 16).astype(np.uint16), B * T * H).reshape(B, T, H).copy()
    y = np.empty_like(x)
    per = []
    for _ in range(n_layers):
        kc = np.zeros((T + 2, B, heads, d), np.uint16)
        vc = np.zeros_like(kc)
        t0 = time.time()
        rc = L.lia_host_layer_forward(ctypes.byref(model.desc), ctypes.byref(arr), x.ctypes.data, y.ctypes.data, kc.ctypes.data, vc.ctypes.data,
                                      T + 2, B, B, T, 0, 0, threads)
        per.append(time.time() - t0)
        if rc != 0:
            raise RuntimeError(f"lia_host_layer_forward: {L.lia_last_error()}")
        x, y = y, x
    flops = 2.0 * B * T * (4.0 * H * H + 2.0 * H * F)
    return {"prefill_ms": 1e3 * min(per) * shape.layers, "kind": f"product host path, {n_layers} layers measured",
            "layer_s": per, "layer_tflops": flops / min(per) / 1e12, "threads": threads,
            "sample": f"lia_host_layer_forward(policy 1) on B={B} x T={T} rows (M={B * T}: the M > 256 host kernel), {n_layers} layers of the "
                      f"{shape.name} shape back to back, best layer x {shape.layers}; embeddings / lm_head excluded"}


def parity_sample(sched, model, shape, B, T, threads):
    """The oracle as CHECKER at the benchmark's own shape: one decode step (S = T + 1) of ONE layer of the model's shape through
    lia_layer_forward (policy 2: GPU linears, host attention over a host cache -- the headline's decode policy) and through
    oracle.layer_forward (fp32 FMA mode) on the SAME tensors.  Reported, not asserted (tests/test_gpu_fullsize_oracle.py asserts)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import torch
    import lia_oracle as orc
    from lia_amd import _native as N, ops
    from lia_amd.model import draw_layer
    orc.lib().lia_oracle_set_threads(threads)
    orc.lib().lia_oracle_set_fast(0)
    H, F, heads = shape.hidden, shape.ffn, shape.heads
    d = H // heads
    bits = lambda t: t.detach().cpu().contiguous().view(torch.int16).numpy().view(np.uint16)  # noqa: E731
    flat = draw_layer(shape, model.offsets, model.layer_bytes, li=7, seed=123)
    g = torch.Generator(device="cuda").manual_seed(5)
    rnd = lambda *shp: torch.randn(*shp, generator=g, device="cuda").to(torch.bfloat16)  # noqa: E731
    x, kc, vc = rnd(B, 1, H), rnd(T + 2, B, heads, d), rnd(T + 2, B, heads, d)
    torch.cuda.synchronize()
    host = bits(flat)
    dims = {"q_w": (H, H), "k_w": (H, H), "v_w": (H, H), "out_w": (H, H), "fc1_w": (F, H), "fc2_w": (H, F)}
    W = {}
    for i, n in enumerate(ops.LAYER_TENSORS):
        shp = dims.get(n, (F,) if n == "fc1_b" else (H,))
        W[n] = host[model.offsets[i] // 2: model.offsets[i] // 2 + int(np.prod(shp))].reshape(shp)
    hk, hv = kc.cpu().pin_memory(), vc.cpu().pin_memory()
    okc, ovc = bits(hk).copy(), bits(hv).copy()
    kv = N.KV(hk.data_ptr(), hv.data_ptr(), T + 2, B, 0)
    y = torch.empty_like(x)
    sched.ctx.layer_forward(model.desc, 2, ops.weight_ptr_array(flat.data_ptr(), model.offsets), x, y, kv, B, 1, T)
    sched.ctx.synchronize()
    t0 = time.time()
    ref = orc.layer_forward(2, W, bits(x), okc, ovc, T, heads)
    oracle_s = time.time() - t0
    got = bits(y)
    f32 = lambda b: (b.astype(np.uint32) << 16).view(np.float32)  # noqa: E731
    a, b = f32(got), f32(ref)
    err = np.abs(a - b)
    q = 2.0 ** (np.floor(np.log2(float(np.abs(b).max()))) - 7)
    return {"what": f"one {shape.name}-shaped layer, decode step B={B} S={T + 1}, policy 2: lia_layer_forward vs oracle.layer_forward (fp32 FMA) on the same tensors",
            "max_abs": float(err.max()), "frac_bit_identical": float((got == ref).mean()), "max_abs_ref": float(np.abs(b).max()),
            "bf16_quantum_at_max_ref": q, "max_err_in_quanta": float(err.max() / q), "frac_within_one_quantum": float((err <= q).mean()),
            "new_kv_row_frac_bit_identical": float(((bits(hk)[T] == okc[T]).mean() + (bits(hv)[T] == ovc[T]).mean()) / 2),
            "oracle_layer_s": oracle_s,
            "note": "per-op identity is >= 99.9 % (tests/test_gpu_fullsize_oracle.py); a whole layer amplifies each op's one-ulp flips "
                    "by ~2*sqrt(p) per GEMM at K >= 7168, hence the lower whole-layer identity rate with a bounded error"}


def wire_stats(model, n_gpu):
    """bits per bf16 value the streamed layers ship, layer by layer: a layer the lossless packing does not shrink (or whose values fall
    outside the format's window too often) is pinned raw by itself -- LayerStore._encode_packed -- and counts 16 bits here"""
    per = [16.0 * st.stream_bytes / st.nbytes for st in model.layers[n_gpu:] if st.tier not in ("device", "remote", None) and not st.shard]
    if not per:
        return None
    return {"layers": len(per), "min": min(per), "mean": sum(per) / len(per), "max": max(per),
            "layers_shipped_raw": sum(1 for st in model.layers[n_gpu:] if st.tier not in ("device", "remote", None) and not st.shard and not st.packed)}


def promote_scalars(out):
    """The driver keeps the standard keys of the line only (metric, value, ..., config, roofline, cpu_baseline): the few scalars a
    reader needs from the other objects ride inside `roofline` and `config` as well."""
    r, c = out.get("roofline"), out.get("config")
    if not isinstance(r, dict) or not isinstance(c, dict):
        return out
    dk = r.get("dominant_kernel") if isinstance(r.get("dominant_kernel"), dict) else (r if r.get("kernel") else {})
    scal = {"prefill_ms": out.get("prefill_ms"), "prefill_ms_defer_kv_0": (out.get("prefill_defer_kv_0_leg") or {}).get("prefill_ms"),
            "value_raw_format": out.get("value_raw_format"), "value_cooperative": out.get("value_cooperative"),
            "value_cooperative_kv_in_hbm": out.get("value_cooperative_kv_in_hbm"),
            "auto_plan_policies": ([(out.get("auto_plan") or {}).get("chosen", {}).get(k) for k in ("prefill_policy", "decoding_policy", "cpu_layers")]
                                   if isinstance(out.get("auto_plan"), dict) and "chosen" in out["auto_plan"] else None),
            "auto_plan_measured_tokens_per_s": (out.get("auto_plan") or {}).get("measured_tokens_per_s"),
            "cooperative_converged": [bool(((out.get(k) or {}).get("controller") or {}).get("converged"))
                                      for k in ("cooperative_leg", "cooperative_kv_in_hbm_leg") if isinstance(out.get(k), dict) and "controller" in out[k]] or None,
            "dominant_kernel_frac": dk.get("frac"), "dominant_kernel_avg_launch_us": dk.get("avg_launch_us"),
            "prefill_mfma_frac": (out.get("prefill_detail") or {}).get("mfma_frac"),
            "parity_max_err_in_quanta": (out.get("parity") or {}).get("max_err_in_quanta"),
            "parity_frac_bit_identical": (out.get("parity") or {}).get("frac_bit_identical"),
            "ids_first_divergent_step": {k: v.get("first_divergent_step") for k, v in (out.get("ids_check") or {}).items()} or None,
            "ids_top2_gap_at_divergence": {k: v.get("top2_logit_gap_at_divergence") for k, v in (out.get("ids_check") or {}).items()
                                           if v.get("first_divergent_step") is not None} or None}
    r["scalars"] = {k: v for k, v in scal.items() if v is not None}
    hl = out.get("host_link") or {}
    c["stream_format"] = hl.get("stream_format")
    c["bits_per_value"] = hl.get("bits_per_value_by_layer") or hl.get("bits_per_value")
    return out


def coop_windows_run(generate, model, ids, kwargs, a, B, hostinfo, room):
    """one cooperative leg: a generate() of 2 + --coop-steps + 8 * --coop-windows steps (its own cache size: the headline's
    max_new_tokens may be smaller); the last 8 * windows decode steps are cut into windows of 8 -> (ids, lat, logits, value = the
    MEDIAN window's tokens/s, [per window: ms per step, tokens/s, CFS throttle periods / ms])"""
    W = max(1, min(a.coop_windows, (room - 2 - a.coop_steps) // 8))       # room: positions left behind the prompt
    total = 2 + a.coop_steps + 8 * W
    kw = dict(kwargs, max_new_tokens=total, min_new_tokens=total)
    marks = {}

    def hook(step):
        if step >= total - 8 * W and (step - (total - 8 * W)) % 8 == 0:
            marks[step] = hostinfo.cgroup_cpu_throttle()
    ids_out, lat, logits = generate(model, ids, return_logits=True, step_hook=hook, **kw)
    marks[total] = hostinfo.cgroup_cpu_throttle()
    wins = []
    for w in range(W):
        s0 = total - 8 * (W - w)
        seg = lat[s0:s0 + 8]
        thr0, thr1 = marks.get(s0), marks.get(s0 + 8)
        wins.append({"steps": [s0, s0 + 8], "ms_per_step": 1e3 * sum(seg) / len(seg), "tokens_per_s": B / (sum(seg) / len(seg)),
                     "cpu_throttle": ({"periods": thr1[0] - thr0[0], "throttled_ms": (thr1[1] - thr0[1]) / 1e3} if thr0 and thr1 else None)})
    rates = sorted(w["tokens_per_s"] for w in wins)
    return ids_out, lat, logits, rates[len(rates) // 2], wins


def throttle_delta(before):
    """CFS quota stalls of the container since `before` (hostinfo.cgroup_cpu_throttle()): the host-computed legs run sixteen threads
    against a sixteen-CPU quota, so a box whose neighbours or helper threads push it over shows up here, not in the kernels"""
    from lia_amd import hostinfo
    now = hostinfo.cgroup_cpu_throttle()
    return {"periods": now[0] - before[0], "throttled_ms": (now[1] - before[1]) / 1e3}


def first_divergence(ids_a, ids_b, T, logits_b=None):
    """compare the generated tokens of two legs over their common length -> {"ids_equal", "steps_compared", "first_divergent_step",
    "top2_logit_gap_at_divergence"} (the gap from leg b's logits of that step, row 0)"""
    import numpy as np
    n = min(ids_a.shape[1], ids_b.shape[1]) - T
    a, b = ids_a[:, T:T + n].numpy(), ids_b[:, T:T + n].numpy()
    diff = np.nonzero((a != b).any(axis=0))[0]
    out = {"ids_equal": bool(diff.size == 0), "steps_compared": int(n), "first_divergent_step": None if diff.size == 0 else int(diff[0])}
    if diff.size and logits_b is not None and int(diff[0]) < len(logits_b):
        import torch
        lg = logits_b[int(diff[0])][0].float()
        top = torch.topk(lg, 2).values
        out["top2_logit_gap_at_divergence"] = float(top[0] - top[1])
        out["bf16_quantum_at_top_logit"] = 2.0 ** (math.floor(math.log2(max(abs(float(top[0])), 1e-30))) - 7)
        out["note"] = ("the legs run different arithmetic on purpose (host-computed layers use the CPU policy's fused-bias linears and fp32 attention, "
                       "decoder.py / Krnl.cpp); they may part only where the top two logits are within a few bf16 quanta")
    return out


def pmc_traffic(kernel_substr):
    """HBM bytes per launch of the dominant kernel from the committed PMC pass of this same command
    (the newest profiles/r*_opt30b_bench_pmc_hbm.json, tools/profile_round.sh: separate --pmc FETCH_SIZE / WRITE_SIZE runs, gfx950 x2 fetch correction).
    PMC collection cannot run inside the timed benchmark, so the live line carries the committed measurement and names it."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_opt30b_pmc_hbm.json")) +
                   glob.glob(os.path.join(ROOT, "profiles", "r*_opt30b_bench_pmc_hbm.json")), key=os.path.basename)
    if not files:
        return None, None
    try:
        d = json.load(open(files[-1]))
        hits = [v for k, v in d["kernels"].items() if kernel_substr in k and "traffic_bytes_per_launch" in v]
        calls = sum(v["calls"] for v in hits)
        if calls:
            return sum(v["traffic_bytes_per_launch"] * v["calls"] for v in hits) / calls, os.path.relpath(files[-1], ROOT)
    except (OSError, ValueError, KeyError):
        pass
    return None, None


WATCHDOG_EXIT_CODE = 3
MIN_HOST_THREADS = 4      # a rank whose policy-2 host attention gets fewer threads than this is flagged (per_rank[].host_threads_starved):
                          # the step is then bound by the shared CPU quota, not by a GPU or a link


def watchdog_line(out, progress, timeout_s):
    """the headline line as the watchdog re-prints it: what finished, which leg was running when the time ran out"""
    line = dict(out)
    line.update(progress.get("res", {}))
    line["dp_extra_legs"] = {"timed_out": True, "leg_running": progress.get("current"), "timeout_s": timeout_s,
                             "legs_finished": sorted(k[:-4] for k in progress.get("res", {}) if k.endswith("_leg")),
                             "exit_code": WATCHDOG_EXIT_CODE}
    return line


WATCHDOG_GRACE_S = 2.0


def watchdog_fire(out, progress, rank, timeout_s, _exit=os._exit, _sleep=time.sleep):
    # a thread, not SIGALRM: the main thread of a hung rank sits inside a C call (an RCCL wait, a stream synchronize) and would never
    # reach a Python signal handler.  No restart, no exec: the process has touched the GPU -- it reports and ends.
    sys.stderr.write(f"bench.py: rank {rank}: the extra data-parallel leg {progress.get('current')!r} exceeded {timeout_s} s; "
                     f"keeping the headline line, exit code {WATCHDOG_EXIT_CODE}\n")
    sys.stderr.flush()
    if rank == 0:
        print(json.dumps(promote_scalars(watchdog_line(out, progress, timeout_s))), flush=True)
    else:
        # the launcher (torch.distributed.run) SIGTERMs every worker ~0.1 s after the first one fails: the other ranks hold on long
        # enough for rank 0 to serialise and flush its line
        _sleep(WATCHDOG_GRACE_S)
    _exit(WATCHDOG_EXIT_CODE)


def dp_extra_legs(a, dist, backend, group, model, sched, shape, ids, gen_kwargs, B, world, rank, n_gpu, fmt, host_threads, dev_index, progress=None,
                  rows_total=None):
    """N > 1 (every rank calls this): value_policy_0_2 (or value_kv_in_hbm when the headline leg ran 0 / 2) -- the same broadcast
    stream with the other cache placement: host attention on each rank's share of the CPU quota vs the KV cache in HBM --,
    value_allgather -- every rank pins 1/N of each streamed layer and reads it over ITS OWN host link, one all-gather per layer over
    xGMI -- and value_allgather_<other placement>, both together.  Short legs (--dp-extra-steps)."""
    import torch
    from lia_amd.generation import generate
    from lia_amd.model import LiaOPTModel
    from lia_amd.scheduler import OffloadScheduler
    dev = "cuda" if backend == "nccl" else "cpu"
    progress = {} if progress is None else progress
    rows_total = B * world if rows_total is None else rows_total
    res = progress.setdefault("res", {})            # (the watchdog reads what has finished, and which leg was running, from here)

    def leg(name, mdl, kwargs):
        t0 = time.time()
        progress["current"] = name
        try:
            generate(mdl, ids, max_steps=2, **kwargs)                          # placement / allocations of this leg, untimed
            sc = mdl._lia_scheduler
            sc.stream_stats(reset=True)
            dist.barrier()
            torch.cuda.synchronize()
            _, lat = generate(mdl, ids, max_steps=2 + a.dp_extra_steps, **kwargs)
            dec = sum(lat[2:]) / len(lat[2:])
            h2d_b, _ = sc.stream_stats()
            v = torch.tensor([dec, lat[0], h2d_b / max(sum(lat), 1e-9) / 1e9], dtype=torch.float64, device=dev)
            allv = [torch.zeros_like(v) for _ in range(world)]
            dist.all_gather(allv, v)
            worst = max(float(x[0]) for x in allv)
            res["value_" + name] = rows_total / worst
            res[name + "_leg"] = {"decode_steps_timed": len(lat[2:]), "ms_per_step": 1e3 * worst, "prefill_ms": 1e3 * max(float(x[1]) for x in allv),
                                  "per_rank_h2d_gbs": [float(x[2]) for x in allv], "leg_s": time.time() - t0}
        except Exception as e:                                                  # (every rank takes the same branch: same shapes, same flags)
            res[name + "_leg"] = {"error": f"{type(e).__name__}: {e}"}
        progress["current"] = None

    # the OTHER cache placement than the headline leg's: with the default N > 1 policies (3 / 3, plan_policies) that is the
    # reference's 0 / 2 with host attention on this rank's share of the CPU quota -- `value_policy_0_2` --, with 0 / 2 named on the
    # command line it is `value_kv_in_hbm`
    headline_kv_in_hbm = (gen_kwargs.get("prefill_policy"), gen_kwargs.get("decoding_policy")) == (3, 3)
    alt_name, alt = ("policy_0_2", dict(prefill_policy=0, decoding_policy=2)) if headline_kv_in_hbm else ("kv_in_hbm", dict(prefill_policy=3, decoding_policy=3))
    leg(alt_name, model, dict(gen_kwargs, **alt))
    if group.mode != "allgather" and world > 1 and a.dp_allgather_legs:
        # every rank needs its own slice of every streamed layer: the root gives its copies up, all ranks draw the (seeded) layers
        # again and pin slice r of G.  The resident layers and the head stay as they are.
        sched.close()
        for st in model.layers[n_gpu:]:
            st.close()
        group.mode = "allgather"
        try:
            m2 = LiaOPTModel.random_init(shape, seed=0, init=a.init, n_gpu_layers=n_gpu, pin_weight=True, host_owner=True, wire=fmt,
                                         shard=(rank, world))
            m2._lia_scheduler = OffloadScheduler(m2, device=dev_index, dp_group=group, wire=fmt)
            m2._lia_scheduler.host_threads = host_threads
            leg("allgather", m2, gen_kwargs)
            # ... and both together: every rank's own link AND no host attention -- the configuration in which neither the root's
            # PCIe link nor the shared CPU quota bounds the step
            leg("allgather_" + alt_name, m2, dict(gen_kwargs, **alt))
            m2._lia_scheduler.close()
            m2.close()
        except Exception as e:
            res["allgather_leg"] = {"error": f"{type(e).__name__}: {e}"}
    return res


def plan_rows(a, rank, world):
    """(rows of this rank, rows of the whole job): --batch is per GPU (weak scaling), --global-batch is split over the ranks with
    the remainder on the first ones (BASELINE config 5: 256 over 8)"""
    from lia_amd import dp
    if a.global_batch:
        if a.global_batch < world:
            raise SystemExit(f"--global-batch {a.global_batch} leaves some of the {world} GPUs without a row")
        lo_, hi_ = dp.shard_rows(a.global_batch, rank, world)
        return hi_ - lo_, a.global_batch
    return a.batch, a.batch * world


def kv_cache_bytes(shape, rows, positions, layers=None):
    """K and V rows of `layers` layers (default: all) for `rows` sequences of `positions` tokens, bf16"""
    return 2 * 2 * positions * rows * shape.hidden * (shape.layers if layers is None else layers)


def plan_policies(a, world, shape, rows, T, new, layer_bytes, hbm_bytes, n_gpu):
    """(prefill policy, decode policy, why) when the command line names none.  One GPU: the reference's 0 / 2 (BASELINE configs[1]).
    N > 1: the ranks of a node share ONE host -- 16 CPUs of quota on the GPU box, i.e. 2 host-attention threads per rank at N = 8,
    ~480 ms of host attention in a 650 ms step -- so the default leg keeps every layer's KV cache in HBM
    (policies 3 / 3) whenever it fits beside the resident layers, the streamer slots + staging and the workspace;
    the 0 / 2 run is then the extra leg `value_policy_0_2`."""
    if a.prefill_policy is not None or a.decoding_policy is not None:
        return (0 if a.prefill_policy is None else a.prefill_policy), (2 if a.decoding_policy is None else a.decoding_policy), "named on the command line"
    if world <= 1:
        return 0, 2, "one GPU: the reference's prefill 0 / decode 2"
    need = kv_cache_bytes(shape, rows, T + new) + (n_gpu + 4 + 4) * layer_bytes + 2 * kv_cache_bytes(shape, rows, T, 1) + (6 << 30)
    if need <= 0.85 * hbm_bytes:
        return 3, 3, f"N = {world}: KV of every layer in HBM ({need / 2**30:.1f} GiB of {hbm_bytes / 2**30:.0f} GiB per rank), no host attention on the shared CPU quota"
    return 0, 2, f"N = {world}: the KV cache does not fit HBM beside the layers ({need / 2**30:.1f} GiB of {hbm_bytes / 2**30:.0f} GiB): host attention"


def reduce_over_ranks(dist, backend, rank, world, elapsed, prefill_ms, dec_mean_s, host_threads, host_attn_ms_step, my_h2d_gbs, B):
    """MAX over ranks of the three times + every rank's host-thread / link figures, through REAL collectives of the communicator
    the run used: (elapsed, prefill_ms, dec_mean_s, rccl_ranks, per_rank)"""
    import torch
    per_rank = [{"rank": 0, "host_attention_threads": host_threads, "host_attention_ms_per_step": host_attn_ms_step,
                 "h2d_gbs": my_h2d_gbs, "rows": B, "host_threads_starved": host_threads < MIN_HOST_THREADS}]
    if dist is None:
        return elapsed, prefill_ms, dec_mean_s, 1, per_rank
    dev = "cuda" if backend == "nccl" else "cpu"
    tmax = torch.tensor([elapsed, prefill_ms, dec_mean_s], dtype=torch.float64, device=dev)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    mine = torch.tensor([float(rank), float(host_threads), host_attn_ms_step, my_h2d_gbs, float(B)], dtype=torch.float64, device=dev)
    allr = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allr, mine)
    per_rank = [{"rank": int(v[0]), "host_attention_threads": int(v[1]), "host_attention_ms_per_step": float(v[2]),
                 "h2d_gbs": float(v[3]), "rows": int(v[4]), "host_threads_starved": int(v[1]) < MIN_HOST_THREADS} for v in allr]
    return float(tmax[0]), float(tmax[1]), float(tmax[2]), dist.get_world_size(), per_rank


def dp_predicted(a, shape, world, group_mode, T, new):
    """the planner's prediction for THIS line (lia_amd.planner.predict_dp, committed as a table in DESIGN.md section 6 before any
    multi-GPU hardware ran the code): ms per step, the binding resource and every term, for the line's N, rows, mode and both
    cache placements -- so that a measured N > 1 line can be read against what was expected of it"""
    try:
        from lia_amd import planner
        box = planner.Box(wire_ratio={"raw": 1.0, "pack10": 0.675}.get(a.stream_format or "pack10", 0.675))
        rows = -(-a.global_batch // world) if a.global_batch else a.batch
        out = {}
        for pol in (3, 2):
            ms, bound, terms = planner.predict_dp(shape, rows, T, new, a.gpu_percentage, world, group_mode or "broadcast", pol, box)
            out["3/3" if pol == 3 else "0/2"] = {"ms_per_step": round(ms, 1), "tokens_per_s": round(1e3 * rows * world / ms, 1), "bound_by": bound,
                                                   **{k: (round(v, 1) if isinstance(v, float) else v) for k, v in terms.items()}}
        out["assumptions"] = (f"link {box.link_gbs} GB/s, xGMI link {planner.XGMI_LINK_GBS} GB/s x ring efficiency {planner.RCCL_RING_EFF} (assumed), "
                              f"decode GEMM {box.hbm_gbs} GB/s, host attention {box.host_gbs_per_thread} GB/s per thread on 16 / N threads")
        return out
    except Exception as e:      # noqa: BLE001  (a prediction must never cost the measured line)
        return {"error": f"{type(e).__name__}: {e}"}


def dp_config_fields(a, shape, rows_total, world, group_mode, host_threads, policies):
    from lia_amd import dp
    return {"global_batch": rows_total, "rows_per_rank": [dp.shard_rows(rows_total, r, world)[1] - dp.shard_rows(rows_total, r, world)[0] for r in range(world)],
            "parallelism": (f"dp{world} batch-shard, {group_mode} weight stream" if world > 1 else "single GPU"),
            "host_attention_threads": host_threads, "policies": {"prefill": policies[0], "decode": policies[1], "why": policies[2]}}


def dp_line_selftest(a):
    """CPU-only (gloo) body behind --selftest-dp-line: everything the N > 1 line says ABOUT the job split -- row plan, default
    policies, host threads per rank, the MAX / all-gather reductions, config.rows_per_rank -- through the same functions main()
    uses, with synthetic per-rank timings (rank r: 10 + r ms per step).  No GPU, no model: tests/test_bench_launcher.py."""
    import torch.distributed as dist
    from lia_amd import hostinfo
    from lia_amd.model import resolve_shape
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    shape = resolve_shape(a.model)
    B, rows_total = plan_rows(a, rank, world)
    T, new = a.prompt, 1 + a.warmup + a.steps
    n_gpu = int(shape.layers * a.gpu_percentage / 100)
    pol = plan_policies(a, world, shape, B, T, new, 2 * (4 * shape.hidden ** 2 + 2 * shape.hidden * shape.ffn), 288 * 2 ** 30, n_gpu)
    host_threads = a.host_threads or hostinfo.default_host_threads(world)
    step_s = (10.0 + rank) * 1e-3
    elapsed, prefill_ms, dec_mean_s, ranks, per_rank = reduce_over_ranks(dist, "gloo", rank, world, step_s * a.steps, 100.0 + rank, step_s, host_threads,
                                                                          1.5 * rank, 50.0 if rank == 0 else 0.0, B)
    dist.barrier()
    if rank == 0:
        cfg = {"workload": f"{shape.name} shape, dp line selftest"}
        cfg.update(dp_config_fields(a, shape, rows_total, world, "broadcast", host_threads, pol))
        cfg["predicted"] = dp_predicted(a, shape, world, "broadcast", T, new)
        print(json.dumps({"metric": "dp line selftest", "value": rows_total * a.steps / elapsed, "unit": "tokens/s", "n_gpus": world, "steps": a.steps,
                          "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps, "scaling": "strong" if a.global_batch else "weak",
                          "prefill_ms": prefill_ms, "collective_ranks": ranks, "rccl_ranks": ranks, "collective_backend": "gloo", "per_rank": per_rank, "config": cfg,
                          "dp_line_selftest": True}), flush=True)
    dist.destroy_process_group()
    return 0


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    a = build_parser().parse_args(argv)
    if a.gpus > 1 and "RANK" not in os.environ:
        return self_launch(a.gpus, argv)
    if a.selftest_launcher:
        return launcher_selftest()
    if a.selftest_dp_line:
        return dp_line_selftest(a)

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus > 1 and world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    # --dp-same-gpu (validation only, with --dp-backend gloo): several ranks share one GPU, so the whole
    # batch-shard path (remote tiers, chunked broadcast into staging, decode on non-root ranks) runs on a 1-GPU box
    dev_index = local_rank % max(1, torch.cuda.device_count()) if a.dp_same_gpu else local_rank
    backend = a.dp_backend
    torch.cuda.set_device(dev_index)
    # the reference pins its CPU work with `numactl -m 0 -C 0-39` (README.md:78); here: the cores of the GPU's NUMA node, where
    # the pinned weights and KV caches live (LIA_PIN_NODE=<n> overrides, -1 = no pinning).  Must precede the first OpenMP team.
    from lia_amd import hostinfo
    pin_node = hostinfo.pin_node(dev_index)
    pinned_cpus = hostinfo.pin_to_node(pin_node) if pin_node >= 0 else 0
    dist = None
    force_dp = a.force_dp                                 # exercise the broadcast path on a single GPU (world 1)
    if world > 1 or force_dp:
        import torch.distributed as dist
        if "RANK" not in os.environ:
            os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)

    from lia_amd.model import LiaOPTModel, resolve_shape
    from lia_amd.generation import generate
    from lia_amd.scheduler import OffloadScheduler, default_stream_format
    from lia_amd import dp
    if a.stream_format is None:
        a.stream_format = default_stream_format()

    is_llama = "llama" in a.model.lower()
    if is_llama:
        from lia_amd.llama import LiaLlamaModel, LlamaScheduler, resolve_llama_shape
        shape = resolve_llama_shape(a.model)
    else:
        shape = resolve_shape(a.model)
    B, rows_total = plan_rows(a, rank, world)
    T = a.prompt
    new = 1 + a.warmup + a.steps
    if T + new > shape.max_pos:
        raise SystemExit("prompt + steps exceeds max positions")
    n_gpu = shape.layers if (is_llama and a.gpu_percentage >= 100) else int(shape.layers * a.gpu_percentage / 100)
    if a.bracket_stride <= 0:
        a.bracket_stride = 32 if n_gpu >= shape.layers else 8       # both co-prime to the 193 / 129 launches of an OPT-30B / Llama-3-8B step
    host_threads = a.host_threads or hostinfo.default_host_threads(world)
    layer_bytes_est = 2 * (4 * shape.hidden ** 2 + 2 * shape.hidden * shape.ffn) if not is_llama else 0
    # --dp-same-gpu (dry runs): the ranks share ONE card, so each may plan with its share of the HBM only
    ranks_per_device = max(1, -(-world // max(1, torch.cuda.device_count()))) if a.dp_same_gpu else 1
    hbm_per_rank = torch.cuda.get_device_properties(dev_index).total_memory // ranks_per_device
    if is_llama:
        # a Llama has no LIA policy (decoder.py:121-169): policies named on the command line are left as they are and not reported
        policies = (a.prefill_policy, a.decoding_policy, "llama: no LIA policies (every layer on the GPU)")
    else:
        policies = plan_policies(a, world, shape, B, T, new, layer_bytes_est, hbm_per_rank, n_gpu)
        a.prefill_policy, a.decoding_policy = policies[0], policies[1]
    # --cpu-layers -1: the scheduler's online controller picks the count from the measured decode steps, starting on the count this
    # box converged on last time (scheduler.CoopStore) or else on planner.plan_cpu_layers' estimate -- no explicit start from here
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
    fmt = WIRE[a.stream_format]
    if is_llama:
        model = LiaLlamaModel.random_init(shape, seed=0, n_gpu_layers=n_gpu)
        sched = LlamaScheduler(model, device=dev_index)
    else:
        model = LiaOPTModel.random_init(shape, seed=0, init=a.init, n_gpu_layers=n_gpu, pin_weight=True, enable_cxl=a.enable_cxl,
                                        host_owner=(group is None or group.is_root or group.mode == "allgather"), wire=fmt,
                                        shard=((rank, world) if (group is not None and world > 1 and group.mode == "allgather") else None),
                                        raw_layers=(OffloadScheduler.cpu_layer_set(n_gpu, shape.layers, a.cpu_layers)
                                                    if (a.cpu_layers > 0 and a.decoding_policy in (2, 3) and group is None) else ()))
        sched = OffloadScheduler(model, device=dev_index, dp_group=group, wire=fmt)
        sched.host_threads = host_threads
    model._lia_scheduler = sched                         # generate() drives this scheduler
    g = torch.Generator().manual_seed(0)
    row = torch.randint(4, shape.vocab, (T,), generator=g, dtype=torch.int64)
    row[0] = 2
    ids = row[None, :].repeat(B, 1)                      # identical rows, run_generation.py:285 (every rank: its B rows)
    build_s = time.time() - t_build

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    gen_kwargs = dict(do_sample=False, num_beams=1, max_new_tokens=new, min_new_tokens=new, token_latency=True, **flags)
    # untimed shake-out (the harness's warm-up iteration): allocations (pinned KV, workspace, slots), page-in, clocks
    generate(model, ids, max_steps=3, **gen_kwargs)

    st = {}

    def hook(step):
        if step == 0:
            sched.stream_stats(reset=True)
            sched.ctx.prof_start(4096)
            sync()
        if step == 1:
            st["prof_prefill"] = sched.ctx.prof_stop()
            st["pre_h2d"] = sched.stream_stats()
        if step == 1 + a.warmup:                         # exactly --steps decode steps follow
            sched.stream_stats(reset=True)
            if not is_llama:
                sched.decode_stats(reset=True, block=False)      # (never drain the prefetched layers' decodes in front of the first timed step)
            sched.ctx.prof_start(16384, stride=a.bracket_stride)
            sync()
            st["thr0"] = hostinfo.cgroup_cpu_throttle()
            st["t0"] = time.time()

    out_ids, lat = generate(model, ids, step_hook=hook, **gen_kwargs)
    sync()
    elapsed = time.time() - st["t0"]
    thr1 = hostinfo.cgroup_cpu_throttle()
    prof = sched.ctx.prof_stop()
    h2d_bytes, h2d_ms = sched.stream_stats()
    wire_dec = sched.decode_stats() if not is_llama else {"launches": 0, "ms": 0.0, "bytes_in": 0.0, "bytes_out": 0.0}
    prof_prefill, (pre_h2d_bytes, pre_h2d_ms) = st["prof_prefill"], st["pre_h2d"]
    assert len(lat) == new and out_ids.shape == (B, T + new)
    prefill_ms = 1e3 * lat[0]                             # run_generation.py:345: first-token latency
    dec_mean_s = sum(lat[1:]) / len(lat[1:])              # :346-349 "Average 2... latency"
    timed = sorted(lat[1 + a.warmup:])
    host_attn_ms_step = prof.get("host_attention_ms", 0.0) / max(1, a.steps)

    my_h2d_gbs = h2d_bytes / (elapsed * 1e9)
    elapsed, prefill_ms, dec_mean_s, rccl_ranks, per_rank = reduce_over_ranks(dist, backend, rank, world, elapsed, prefill_ms, dec_mean_s, host_threads,
                                                                              host_attn_ms_step, my_h2d_gbs, B)

    out = None
    if rank == 0:
        tokens = rows_total * a.steps
        sk_n = max(1, prof["skinny_launches"])
        sk_raw_ms = prof["skinny_ms"]
        # the HIP-event bracket reads its own cost too (an empty bracket on the same stream, measured by lia_prof_stop);
        # rocprofv3's kernel durations (profiles/) carry no such term, so it is taken out before dividing
        sk_ms = max(1e-9, sk_raw_ms - sk_n * prof.get("empty_bracket_ms", 0.0))
        achieved = prof["skinny_bytes"] / (sk_ms * 1e-3) / 1e9 if sk_ms > 0 else 0.0
        traffic, traffic_src = (pmc_traffic("lia_gemm_skinny2_kernel<4") if (a.model == "opt-30b" and B == 64) else (None, None))
        headline = (a.model == "opt-30b" and rows_total == 64 and T == 256 and a.gpu_percentage == 10 and not a.cpu_layers)
        config5 = (a.model == "opt-30b" and a.global_batch == 256 and T == 256 and a.gpu_percentage == 10 and world > 1)
        streamed = (not is_llama) and n_gpu < shape.layers
        link_gbs = h2d_bytes / (elapsed * 1e9)
        kvd = dict(getattr(sched, "kv_delivery", None) or {})
        kvd.pop("_issued_at", None)
        wire_bytes = float(sum(s.stream_bytes for s in model.layers[n_gpu:] if s.tier not in ("device", "remote", None))) if not is_llama else 0.0
        raw_bytes = float(sum(s.nbytes for s in model.layers[n_gpu:] if s.tier not in ("device", "remote", None))) if not is_llama else 0.0
        out = {
            "metric": "decode tokens/s (+ prefill ms), OPT-30B bs=64 in256/out32 gpu%=10" if headline
                      else (f"decode tokens/s (+ prefill ms), OPT-30B bs=256 in256/out32 gpu%=10 batch-sharded (BASELINE config 5), policies {a.prefill_policy}/{a.decoding_policy}" if config5
                            else f"decode tokens/s (+ prefill ms), {a.model} bs={rows_total} in{T} gpu%={a.gpu_percentage}"),
            "value": tokens / elapsed, "unit": "tokens/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * elapsed / a.steps, "higher_is_better": True, "scaling": "strong" if a.global_batch else "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"{shape.name} shape (random-init {'U[0,1)' if a.init == 'uniform01' else 'trained-like (per-tensor scales, outlier channels)' if a.init == 'trained-like' else 'N(0,0.02)'}), batch {B if rows_total == B * world else 'ceil(' + str(rows_total) + '/' + str(world) + ')'}/GPU identical rows, "
                                   f"prompt {T}, {new} new tokens, gpu%={a.gpu_percentage} ({n_gpu} resident + {shape.layers - n_gpu} streamed layers), "
                                   f"{'' if is_llama else f'prefill policy {a.prefill_policy}, decode policy {a.decoding_policy}, '}pin-weight{', enable-cxl nodes ' + str(a.cxl_nodes) if a.enable_cxl else ''}, "
                                   f"num-minibatch {a.num_minibatch}{(', ' + (str(a.cpu_layers) if a.cpu_layers > 0 else 'an online-chosen number of') + ' decode layers on the host cores') if a.cpu_layers else ''}",
                       **dp_config_fields(a, shape, rows_total, world, (group.mode if group is not None else None), host_threads, policies),
                       **({"predicted": dp_predicted(a, shape, world, (group.mode if group is not None else None), T, new)} if (world > 1 and not is_llama) else {}),
                       "prompt_len": T, "new_tokens": new, "new_tokens_requested": 32 if not is_llama else 128,
                       "new_tokens_note": f"the configuration asks for {32 if not is_llama else 128} new tokens; this run generated 1 + warmup + steps = {new} "
                                          f"(cache sized for {T + new} positions), so the timed decode steps run at S = {T + 1 + a.warmup}..{T + new - 1}",
                       "baseline_config": ("configs[1]" if headline else "configs[4]" if config5 else None),
                       "host_numa_node": pin_node if pinned_cpus else None},
            "prefill_ms": prefill_ms,
            "prefill_ms_note": ("first-token latency = latency_list[0] (run_generation.py:345).  With the policy-0 prefill the K/V rows of the streamed "
                                "layers are parked in HBM and delivered to the host caches AFTER the first token (kv_delivery below; LIA_DEFER_KV=0 "
                                "delivers beside the prefill as the reference's store_cache does, modeling_opt.py:334-345), so unlike the reference's "
                                "number this one does not contain the D2H of the cache; prefill_ms_incl_kv_delivery does") if (streamed and kvd.get("bytes")) else None,
            "kv_delivery": ({"deferred": True, "bytes": kvd.get("bytes"), "device_ms": kvd.get("device_ms"), "gbs": ((kvd.get("bytes") or 0) / kvd["device_ms"] / 1e6) if kvd.get("device_ms") else None,
                             "host_wait_ms_in_first_decode_step": kvd.get("host_wait_ms"), "first_decode_step_ms": 1e3 * lat[1] if len(lat) > 1 else None}
                            if (streamed and kvd.get("bytes")) else {"deferred": False}),
            "prefill_ms_incl_kv_delivery": (prefill_ms + (kvd.get("device_ms") or 0.0)) if (streamed and kvd.get("bytes")) else prefill_ms,
            "prefill_ms_incl_kv_delivery_note": "prefill_ms + the device time of the deferred K/V delivery = when the HOST caches would hold the prompt's K/V had the "
                                                "delivery run right behind the prefill (what the reference's first-token latency contains); here it runs under "
                                                "decode step 1, which it does not slow down (kv_delivery.first_decode_step_ms vs decode_latency_ms.mean)",
            "protocol": {"entry_point": "lia_amd.generation.generate(token_latency=True)", "max_new_tokens": new,
                         "prefill_ms": prefill_ms, "decode_tokens_per_s": rows_total / dec_mean_s,
                         "definition": "prefill = latency_list[0]; decode = batch / mean(latency_list[1:]) (run_generation.py:345-354); "
                                       "`value` brackets the last --steps decode steps with barrier + synchronize"},
            "decode_latency_ms": {"mean": 1e3 * sum(timed) / len(timed), "p90": 1e3 * timed[int(0.9 * (len(timed) - 1))], "max": 1e3 * timed[-1]},
            "roofline": ({"bound": "pcie", "what": "the timed decode step is bound by the host link: every streamed layer crosses it once per step",
                          "achieved": link_gbs, "peak": PCIE_PEAK_GBS, "unit": "GB/s", "frac": link_gbs / PCIE_PEAK_GBS,
                          "traffic": h2d_bytes / a.steps, "traffic_what": "wire bytes per decode step counted by the streamer (lia_stream_stats)",
                          "copy_engine_busy_frac": (h2d_ms * 1e-3) / elapsed, "dominant_kernel": None} if streamed else
                         {"bound": "hbm", "dominant_kernel": None}),
            "dominant_kernel_roofline": {"bound": "hbm", "kernel": "lia_gemm_skinny2_kernel<MT,3,NT,8,RT> (decode linears + lm_head)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src, "launches": prof["skinny_launches"], "bracket_stride": a.bracket_stride, "avg_launch_us": 1e3 * sk_ms / sk_n,
                         "avg_bracket_us_raw": 1e3 * sk_raw_ms / sk_n, "empty_bracket_us": 1e3 * prof.get("empty_bracket_ms", 0.0),
                         "algorithmic_bytes_per_launch": prof["skinny_bytes"] / sk_n,
                         "share_of_step": (sk_ms * a.bracket_stride / a.steps) / (1e3 * elapsed / a.steps)},
            "host_link": {"bound": "pcie", "stream_format": a.stream_format if not is_llama else "raw",
                          "bits_per_value": (16.0 * wire_bytes / raw_bytes) if raw_bytes else None,
                          "bits_per_value_by_layer": wire_stats(model, n_gpu) if not is_llama else None,
                          "weight_bytes_per_step": raw_bytes if not is_llama else None,
                          "achieved": h2d_bytes / (elapsed * 1e9), "peak": PCIE_PEAK_GBS, "unit": "GB/s",
                          "frac": h2d_bytes / (elapsed * 1e9) / PCIE_PEAK_GBS,
                          "copy_engine_busy_frac": (h2d_ms * 1e-3) / elapsed, "bytes_per_step": h2d_bytes / a.steps},
            "prefill_detail": {"gemm_ms": prof_prefill["tiled_ms"], "gemm_launches": prof_prefill["tiled_launches"],
                               "gemm_tflops": prof_prefill["tiled_flops"] / max(prof_prefill["tiled_ms"], 1e-9) / 1e9,
                               "mfma_frac": prof_prefill["tiled_flops"] / max(prof_prefill["tiled_ms"], 1e-9) / 1e9 / MFMA_PEAK_TFLOPS,
                               "h2d_busy_ms": pre_h2d_ms, "h2d_gbs_while_busy": pre_h2d_bytes / max(pre_h2d_ms, 1e-9) / 1e6},
            "collective_ranks": rccl_ranks, "rccl_ranks": rccl_ranks, "rccl_ranks_note": "deprecated name of collective_ranks (it is reported under gloo too); kept for one round",
            "collective_backend": (backend if dist is not None else None), "per_rank": per_rank,
            "host_cpu_throttle": {"periods": thr1[0] - st["thr0"][0], "throttled_ms": (thr1[1] - st["thr0"][1]) / 1e3,
                                  "note": "cgroup CFS quota stalls during the timed decode steps (cpu.stat)"},
            "build_s": build_s,
        }
        if a.cpu_layers < 0 and not is_llama:
            out["cooperative_controller"] = sched.coop_report()
        dk = out.pop("dominant_kernel_roofline")
        dk["ms_per_step"] = sk_ms * a.bracket_stride / a.steps
        if streamed:
            # the kernel with the LARGEST per-step time inside the timed region leads (r04 verdict: the wire-format decode, 44 launches of
            # ~0.5 ms on its own stream, outweighs the decode GEMMs' ~11 ms); the other one is reported beside it
            wk = None
            if wire_dec["launches"]:
                wn = wire_dec["launches"]
                w_ms = max(1e-9, wire_dec["ms"] - wn * prof.get("empty_bracket_ms", 0.0))
                w_bytes = wire_dec["bytes_in"] + wire_dec["bytes_out"]
                w_traffic, w_src = pmc_traffic(f"lia_{a.stream_format}_decode_kernel") if (a.model == "opt-30b" and B == 64) else (None, None)
                wk = {"bound": "hbm", "kernel": f"lia_{a.stream_format}_decode_kernel (wire format -> bf16 layer in the streamer slot, own stream)",
                      "achieved": w_bytes / (w_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": w_bytes / (w_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                      "traffic": w_traffic, "traffic_source": w_src, "launches": wn, "avg_launch_us": 1e3 * w_ms / wn,
                      "avg_bracket_us_raw": 1e3 * wire_dec["ms"] / wn, "algorithmic_bytes_per_launch": w_bytes / wn,
                      "algorithmic_bytes_what": "encoded layer read once + bf16 layer written once",
                      "ms_per_step": w_ms / a.steps, "share_of_step": (w_ms / a.steps) / (1e3 * elapsed / a.steps),
                      "on_critical_path": False}
            if wk is not None and wk["ms_per_step"] >= dk["ms_per_step"]:
                out["roofline"]["dominant_kernel"] = wk
                out["roofline"]["decode_gemm_kernel"] = dk
            else:
                out["roofline"]["dominant_kernel"] = dk
                out["roofline"]["wire_decode_kernel"] = wk
            # what the link DELIVERS in model bytes (every streamed layer's bf16 weights once per step) next to what crosses it
            out["roofline"]["algorithmic_h2d_bytes"] = raw_bytes
            out["roofline"]["algorithmic_h2d_gbs"] = raw_bytes / (elapsed / a.steps) / 1e9
            out["roofline"]["algorithmic_h2d_note"] = ("bf16 weight bytes of the streamed layers per decode step (SURVEY 8d: 54.27 GB for the headline) / step time; "
                                                       "`traffic` / `achieved` are the lossless wire encoding's bytes, which is what the link physically carries")
        else:                       # all-resident: the decode GEMM IS the binding roofline
            out["roofline"] = dict(dk, dominant_kernel=None)

    # ---- second leg: the same streamed layers as RAW bf16 (what the reference ships), same model object, re-tiered ----------
    ids_check = {}
    if rank == 0 and world == 1 and not is_llama and fmt and not a.no_raw_leg and n_gpu < shape.layers:
        sched.wire = 0
        t0 = time.time()
        ids_raw, lat_raw, logits_raw = generate(model, ids, max_steps=2 + a.raw_steps, return_logits=True, **gen_kwargs)
        out["value_raw_format"] = B / (sum(lat_raw[2:]) / len(lat_raw[2:]))
        out["raw_format_leg"] = {"stream_format": "raw", "decode_steps_timed": len(lat_raw[2:]), "prefill_ms": 1e3 * lat_raw[0],
                                 "ms_per_step": 1e3 * sum(lat_raw[2:]) / len(lat_raw[2:]), "retier_and_run_s": time.time() - t0,
                                 "note": "same model object re-placed from the packed wire format to raw bf16; first decode step untimed"}
        ids_check[f"{a.stream_format}_vs_raw_wire"] = first_divergence(out_ids, ids_raw, T, logits_raw)
        del logits_raw
        sched.wire = fmt

    # ---- the prefill with the reference's K/V semantics: store_cache beside the prefill (modeling_opt.py:334-345), not deferred -----
    if rank == 0 and world == 1 and not is_llama and not a.no_defer_kv_leg and n_gpu < shape.layers and a.prefill_policy == 0 \
            and getattr(sched, "defer_kv", False):
        try:
            sched.defer_kv = False
            pre = []
            for _ in range(2):                                                  # (the first one re-sizes nothing, but warms the path)
                _, lat_nd = generate(model, ids, max_steps=1, **gen_kwargs)
                sync()
                pre.append(1e3 * lat_nd[0])
            out["prefill_ms_defer_kv_0"] = min(pre)
            out["prefill_defer_kv_0_leg"] = {"prefill_ms": min(pre), "prefill_ms_runs": pre,
                                             "what": "first-token latency with LIA_DEFER_KV=0: the streamed layers' K/V rows go to the host caches "
                                                     "beside the prefill's weight stream (the reference's store_cache), so this number contains the D2H "
                                                     "of the cache like the reference's does"}
        except Exception as e:
            out["prefill_defer_kv_0_leg"] = {"error": f"{type(e).__name__}: {e}"}
        finally:
            sched.defer_kv = True

    # ---- build-defined cooperative split, beside (never instead of) the headline: the planner's host-computed layer count ------
    if rank == 0 and world == 1 and not is_llama and not a.no_cooperative_leg and not a.cpu_layers and a.decoding_policy == 2 \
            and n_gpu < shape.layers:
        try:
            from lia_amd import planner
            t0, thr_a = time.time(), hostinfo.cgroup_cpu_throttle()
            c0, _ = planner.plan_cpu_layers(shape, B, T, new, a.gpu_percentage,
                                            planner.Box(host_threads=host_threads,
                                                        wire_ratio={"raw": 1.0, "pack10": 0.675}[a.stream_format]))
            # -1: the scheduler's online controller.  No explicit start: it begins on the count this box converged on last time
            # (scheduler.CoopStore, +-1 probes only) when there is one, else on the same plan as c0
            coop_kwargs = dict(gen_kwargs, cpu_layers=-1)
            ids_coop, lat_coop, logits_coop, v_med, wins = coop_windows_run(generate, model, ids, coop_kwargs, a, B, hostinfo, shape.max_pos - T)
            out["value_cooperative"] = v_med
            out["cooperative_leg"] = {"planned_host_layers": c0, "controller": sched.coop_report(), "decode_steps": len(lat_coop) - 1,
                                      "value_is": f"median of {len(wins)} windows of 8 decode steps behind the controller's search", "windows": wins,
                                      "ms_per_step": 1e3 * B / v_med, "leg_s": time.time() - t0,
                                      "cpu_throttle": throttle_delta(thr_a), "host_team": sched.host_team_report(),
                                      "note": "decode layers computed on the host cores never cross the link (build-defined, SURVEY 8 f-3); "
                                              "the count is adjusted online from the measured copy-engine idle time"}
            ids_check["cooperative_vs_headline"] = first_divergence(out_ids, ids_coop, T, logits_coop)
            del logits_coop
        except Exception as e:
            out["cooperative_leg"] = {"error": f"{type(e).__name__}: {e}"}
        if not a.no_cooperative_kv_leg and "error" not in out["cooperative_leg"]:
            # the same split with the other streamed layers' KV cache in HBM (policies 3/3, build-defined): no host attention for the
            # GPU-computed layers; a candidate layer's cache follows it between HBM and the host (KVState.move_cache)
            try:
                t0, thr_a = time.time(), hostinfo.cgroup_cpu_throttle()
                c3, _ = planner.plan_cpu_layers(shape, B, T, new, a.gpu_percentage,
                                                planner.Box(host_threads=host_threads,
                                                            wire_ratio={"raw": 1.0, "pack10": 0.675}[a.stream_format]),
                                                kv_in_hbm=True)
                kv_kwargs = dict(gen_kwargs, prefill_policy=3, decoding_policy=3, cpu_layers=-1)
                ids_kv, lat_kv, logits_kv, v_med, wins = coop_windows_run(generate, model, ids, kv_kwargs, a, B, hostinfo, shape.max_pos - T)
                out["value_cooperative_kv_in_hbm"] = v_med
                out["cooperative_kv_in_hbm_leg"] = {"planned_host_layers": c3, "controller": sched.coop_report(), "decode_steps": len(lat_kv) - 1,
                                                    "value_is": f"median of {len(wins)} windows of 8 decode steps behind the controller's search", "windows": wins,
                                                    "ms_per_step": 1e3 * B / v_med,
                                                    "kv_moved_bytes": sched.kv_moved_bytes, "leg_s": time.time() - t0,
                                                    "cpu_throttle": throttle_delta(thr_a), "host_team": sched.host_team_report()}
                ids_check["cooperative_kv_in_hbm_vs_headline"] = first_divergence(out_ids, ids_kv, T, logits_kv)
                del logits_kv
            except Exception as e:
                out["cooperative_kv_in_hbm_leg"] = {"error": f"{type(e).__name__}: {e}"}

    # ---- CPU baseline: the reference's policy 1 ("compute everything on CPU", IPEX/AMX there) on this box's host cores ----------
    if rank == 0 and world == 1 and not is_llama and not a.no_cpu_baseline:
        cpu_kwargs = dict(gen_kwargs, prefill_policy=0, decoding_policy=1, gpu_percentage=0)
        cpu_kwargs.pop("cpu_layers", None)
        product = None
        try:
            t0, thr_a = time.time(), hostinfo.cgroup_cpu_throttle()
            ids_cpu, lat_cpu, logits_cpu = generate(model, ids, max_steps=2 + a.cpu_steps, return_logits=True, **cpu_kwargs)
            product = {"decode_tokens_per_s": B / (sum(lat_cpu[2:]) / len(lat_cpu[2:])), "decode_steps_timed": len(lat_cpu[2:]),
                       "ms_per_step": 1e3 * sum(lat_cpu[2:]) / len(lat_cpu[2:]), "leg_s": time.time() - t0, "cpu_throttle": throttle_delta(thr_a),
                       "sample": f"generate(prefill_policy=0, decoding_policy=1, gpu_percentage=0): {len(lat_cpu[2:])} full decode steps, every one of the "
                                 f"{shape.layers} layers on the host cores (lia_host_layer_forward: AVX-512-BF16 linears + fp32 attention over the host "
                                 "KV cache, weights read raw from pinned memory); embeddings / final LN / lm_head stay on the GPU; the prefill is the GPU's"}
            ids_check["host_policy1_vs_headline"] = first_divergence(out_ids, ids_cpu, T, logits_cpu)
            del logits_cpu
        except Exception as e:          # the oracle sample below still gives a baseline
            product = {"error": f"{type(e).__name__}: {e}"}
        orc = cpu_oracle_sample(shape, B, T, host_threads)
        try:
            cpu_pre = cpu_product_prefill_sample(model, shape, B, T, host_threads, a.cpu_prefill_layers) if a.cpu_prefill_layers > 0 else None
        except Exception as e:
            cpu_pre = {"error": f"{type(e).__name__}: {e}"}
        # r05 verdict, weak item 8: the leg above keeps embeddings / final LN / lm_head on the GPU, the reference's policy 1 does not.
        # The head is measured on the host cores and ADDED to every step: cpu_baseline.value is a full-CPU decode step.
        try:
            head = cpu_host_head_sample(model, shape, B, host_threads)
            if "ms_per_step" in product:
                product["decode_tokens_per_s_gpu_head"] = product["decode_tokens_per_s"]
                product["ms_per_step_gpu_head"] = product["ms_per_step"]
                product["ms_per_step"] = product["ms_per_step"] + head["ms_per_step"]
                product["decode_tokens_per_s"] = 1e3 * B / product["ms_per_step"]
                product["host_head"] = head
                product["sample"] = product["sample"].replace("embeddings / final LN / lm_head stay on the GPU", "embeddings / final LN / lm_head measured on the host cores "
                                                              "beside it (host_head) and added to every step")
        except Exception as e:
            product["host_head"] = {"error": f"{type(e).__name__}: {e}", "note": "value is host layers + GPU head"}
        best = max(orc["decode_tokens_per_s"], product.get("decode_tokens_per_s", 0.0))
        out["cpu_baseline"] = {"value": best, "unit": "tokens/s", "cores": ((sched.host_team_report() or {}).get("threads", host_threads) if best == product.get("decode_tokens_per_s") else host_threads), "kind": "port",
                               "implementation": ("product host path through generate() + the head (embeddings, final LN, lm_head, argmax) on the host cores: a full-CPU step"
                                                  if best == product.get("decode_tokens_per_s") else "oracle restatement, one-layer sample"),
                               "sample": product.get("sample", orc["sample"]),
                               "product_host_path": product, "oracle_port": orc,
                               "prefill_ms": (cpu_pre or {}).get("prefill_ms", orc["prefill_ms"]),
                               "prefill": (cpu_pre if cpu_pre and "prefill_ms" in cpu_pre else dict(cpu_pre or {}, prefill_ms=orc["prefill_ms"], kind="oracle restatement, one-layer B/8 sample scaled")),
                               "cpu": hostinfo.cpu_model(), "isa": hostinfo.isa_flags(), "cpus_usable": hostinfo.usable_cpus()}
        try:
            out["parity"] = parity_sample(sched, model, shape, B, T, host_threads)
        except Exception as e:
            out["parity"] = {"error": f"{type(e).__name__}: {e}"}
    # ---- what `run.py --auto-plan` picks on THIS box for this configuration (r04 verdict, weak item 8: on a box whose CPU baseline beats
    # the reference's hand-picked 0 / 2, the line itself should show the planner choosing the better policy, next to the leg that
    # measured it): the same calibrate() + plan() + plan_cpu_layers() the harness runs, capped at the configured gpu%
    if rank == 0 and world == 1 and not is_llama and not a.no_auto_plan and n_gpu < shape.layers and not a.cpu_layers:
        try:
            from types import SimpleNamespace
            from lia_amd import run_generation
            msgs = []
            ns = SimpleNamespace(plan_hbm_gb=0.0, plan_max_gpu_percentage=a.gpu_percentage, stream_format=a.stream_format, batch_size=B,
                                 input_tokens=T, max_new_tokens=new, model_id=a.model, gpu_percentage=a.gpu_percentage, prefill_policy=a.prefill_policy,
                                 decoding_policy=a.decoding_policy, num_minibatch=a.num_minibatch, pin_weight=True, cpu_layers=0, cpu_layers_start=0)
            t0 = time.time()
            pl = run_generation.auto_plan(ns, out=msgs.append)
            leg = {(3, 3): "value_cooperative_kv_in_hbm", (0, 2): "value_cooperative"}.get((ns.prefill_policy, ns.decoding_policy)) if ns.cpu_layers else None
            measured = out.get(leg) if leg else (out["value"] if (ns.prefill_policy, ns.decoding_policy, ns.gpu_percentage) == (a.prefill_policy, a.decoding_policy, a.gpu_percentage) else None)
            out["auto_plan"] = {"chosen": {"gpu_percentage": ns.gpu_percentage, "prefill_policy": ns.prefill_policy, "decoding_policy": ns.decoding_policy,
                                           "cpu_layers": (f"online from {ns.cpu_layers_start}" if ns.cpu_layers < 0 else ns.cpu_layers), "stream_format": ns.stream_format},
                                "predicted_tokens_per_s": pl.decode_tokens_per_s, "messages": msgs, "plan_s": time.time() - t0,
                                "measured_by_leg": leg or "value", "measured_tokens_per_s": measured,
                                "vs_hand_picked_headline": (measured / out["value"]) if measured else None,
                                "vs_cpu_baseline": (measured / out["cpu_baseline"]["value"]) if (measured and isinstance(out.get("cpu_baseline"), dict)) else None,
                                "what": "the flags `run.py --auto-plan --plan-max-gpu-percentage <gpu%>` sets on this box (planner.calibrate + plan + "
                                        "plan_cpu_layers), and the leg of THIS run that measured them"}
        except Exception as e:
            out["auto_plan"] = {"error": f"{type(e).__name__}: {e}"}

    if rank == 0:
        # whether a cooperative leg started on a count an EARLIER process left in $LIA_STATE_DIR (scheduler.CoopStore keeps and reads
        # it only when that variable is set): a seeded and an unseeded run of the same box are different measurements
        seeded = {k: bool(((out.get(k) or {}).get("controller") or {}).get("seeded_from_store")) for k in ("cooperative_leg", "cooperative_kv_in_hbm_leg")
                  if isinstance(out.get(k), dict) and "controller" in out[k]}
        if seeded:
            out["cooperative_seeded_from_store"] = seeded
            out["cooperative_state_dir"] = os.environ.get("LIA_STATE_DIR")
    if rank == 0 and ids_check:
        out["ids_check"] = ids_check

    # ---- N > 1: two short extra legs so that ONE line shows why the curve bends (same model, same ranks) -------------------
    if dist is not None and (world > 1 or force_dp) and not is_llama and not a.no_dp_extra_legs and n_gpu < shape.layers:
        # The headline of this run is already measured.  It is printed NOW, and again -- extended -- as the last line when the legs
        # are done: should a leg hang in a collective (one rank failing where the others do not), a watchdog ends every rank with
        # exit code 3 (a process killed inside a collective has NOT succeeded) after rank 0 has re-printed the headline line with
        # the name of the leg that hung, so the measured headline is still the last JSON line of the output.
        import threading
        if rank == 0:
            out["dp_extra_legs"] = "pending (this line is re-printed with value_policy_0_2 (or value_kv_in_hbm) -- and value_allgather* under --dp-allgather-legs -- when they finish)"
            print(json.dumps(promote_scalars(out)), flush=True)
        progress = {}
        timer = threading.Timer(a.dp_extra_timeout, watchdog_fire, args=(out, progress, rank, a.dp_extra_timeout))
        timer.daemon = True
        timer.start()
        extra = dp_extra_legs(a, dist, backend, group, model, sched, shape, ids, gen_kwargs, B, world, rank, n_gpu, fmt, host_threads, dev_index,
                              progress=progress, rows_total=rows_total)
        timer.cancel()
        if rank == 0:
            out.pop("dp_extra_legs", None)
            out.update(extra)
    if rank == 0:
        out["host_memory_gib"] = {k: (None if v is None else round(v / 2**30, 2)) for k, v in hostinfo.cgroup_memory().items()}

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
        print(json.dumps(promote_scalars(out)), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
