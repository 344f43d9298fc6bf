"""Benchmark harness: counterpart of llm/single_instance/run_generation.py (reference :60-117 flags,
:179-182 generate kwargs, :285 identical-row batch, :308-335 timing loop, :337-354 summary).

Differences forced by the environment: there is no network, so no HF checkpoint / tokenizer / prompt.json
(tools/env_setup.sh:221 downloads it) -- the prompt is `--input-tokens` synthetic token ids drawn with
seed 0 (row 0 = BOS), and weights are a seeded random init of the named OPT shape unless `-m` points at a
local HF checkpoint directory.  Everything else -- the protocol and the four summary lines -- is the
reference's.
"""
import argparse
import os
import time

import numpy as np
import torch

from .generation import generate
from .model import LiaOPTModel, resolve_shape
from .scheduler import default_stream_format


def build_parser():
    p = argparse.ArgumentParser("Generation script (bf16 path, MI355X)", add_help=True)
    p.add_argument("-m", "--model-id", type=str, default="facebook/opt-30b", help="OPT shape name or local HF checkpoint dir")
    p.add_argument("--dtype", type=str, choices=["bfloat16"], default="bfloat16")
    p.add_argument("--input-tokens", default="32", type=str)
    p.add_argument("--max-new-tokens", default=32, type=int)
    p.add_argument("--greedy", action="store_true")
    p.add_argument("--ipex", action="store_true", help="accepted for command-line compatibility; no effect")
    p.add_argument("--benchmark", action="store_true")
    p.add_argument("--num-iter", default=100, type=int)
    p.add_argument("--num-warmup", default=10, type=int)
    p.add_argument("--batch-size", default=1, type=int)
    p.add_argument("--token-latency", action="store_true")
    p.add_argument("--prompt", default=None, type=str,
                   help="input prompt for self-defined if needed (run_generation.py:87); needs the tokenizer files in the -m directory")
    p.add_argument("--profile", action="store_true",
                   help="one profiled generate() before the timed loop (run_generation.py:103,290-307 runs torch.profiler over five "
                        "there); here the library's own HIP-event brackets: GEMM time / rates per phase, host attention, link traffic")
    # the seven LIA flags, same names / types / defaults as run.py:195-215 and run_generation.py:111-117
    p.add_argument("--prefill-policy", default=1, type=int)
    p.add_argument("--decoding-policy", default=1, type=int)
    p.add_argument("--no-overlap", action="store_true")
    p.add_argument("--pin-weight", action="store_true")
    p.add_argument("--gpu-percentage", default=0, type=int)
    p.add_argument("--num-minibatch", default=1, type=int)
    p.add_argument("--enable-cxl", action="store_true")
    # build-specific
    p.add_argument("--stream-format", default=None, choices=["raw", "pack10"],
                   help="wire format of the pinned streamed layers: a lossless packed format (default pack10, or $LIA_STREAM_FORMAT; same "
                        "results bit for bit, fewer bytes over the host link; layers that do not pack are pinned raw) or raw bf16 (what the "
                        "reference ships); a model directory in the build's packed format defaults to the format on disk")
    p.add_argument("--auto-plan", action="store_true",
                   help="measure this box (lia_amd.planner.calibrate, a few seconds) and let the planner choose --gpu-percentage, "
                        "the policies and --cpu-layers instead of the hand-picked values of llm/scripts/lia_*.sh")
    p.add_argument("--plan-max-gpu-percentage", default=100, type=int,
                   help="--auto-plan what-if: the plan keeps at most this share of the layers resident (the others must stream)")
    p.add_argument("--plan-hbm-gb", default=0.0, type=float, help="--auto-plan what-if: plan as if the GPU had this much HBM")
    p.add_argument("--cpu-layers", default=0, type=int,
                   help="with --decoding-policy 2: this many streamed layers take their decode step on the host cores (policy 1 per layer); "
                        "-1 = chosen online from the measured decode steps (scheduler.CoopController), seeded by the planner")
    p.add_argument("--cpu-layers-start", default=None, type=int, help=argparse.SUPPRESS)
    p.add_argument("--cxl-nodes", default=None, type=str,
                   help="NUMA nodes of the --enable-cxl tier, e.g. 2,3 (the reference hard-codes {2, 3}, lia/cxl/numa_alloc.c:80-81; default: "
                        "$LIA_CXL_NODES, else {2, 3} where both exist, else this box's memory nodes other than the GPU's own)")
    p.add_argument("--result-json", default=None, type=str,
                   help="write one JSON record of the run to this path (tools/run_matrix.py): the summary numbers, the weight stream's "
                        "bytes / copy-engine time over the timed iterations, the library's per-launch brackets of the FIRST (warm-up) "
                        "iteration, the host-memory peak, the planner's pick for the same line; a run the box cannot hold (MemoryError) "
                        "is recorded with its reason instead of a traceback")
    p.add_argument("--seed", default=0, type=int)
    p.add_argument("--init", default="normal", choices=["normal", "uniform01", "trained-like"],
                   help="uniform01 = the reference's dummy-weight recipe (utils/opt-weight-gen.py:61-62)")
    return p


def synthetic_prompt(vocab, n_tokens, batch, seed=0):
    g = torch.Generator().manual_seed(seed)
    row = torch.randint(4, vocab, (n_tokens,), generator=g, dtype=torch.int64)
    row[0] = 2
    return row[None, :].repeat(batch, 1)          # prompt = [prompt] * batch_size  (run_generation.py:285)


TOKENIZER_FILES = ("tokenizer.json", "tokenizer.model", "vocab.json", "tokenizer_config.json")


def choose_cxl_nodes(named=None):
    """The node set of the NUMA / CXL tier.  The reference hard-codes nodes {2, 3} of ITS machine (lia/cxl/numa_alloc.c:80-81); a box
    without them would fail every --enable-cxl line with "Fail to allocate CXL memory!".  Order: --cxl-nodes, $LIA_CXL_NODES (read by
    the library itself: None is returned), {2, 3} where both have memory, else the box's memory nodes other than GPU 0's own (all of
    them on a one- or two-node box)."""
    from . import hostinfo
    from .cxl.numa_alloc import set_cxl_nodes
    if named:
        nodes = [int(v) for v in named.split(",")]
    elif os.environ.get("LIA_CXL_NODES"):
        return None
    else:
        have = hostinfo.numa_nodes()
        if 2 in have and 3 in have:
            nodes = [2, 3]
        else:
            own = hostinfo.gpu_numa_node(0)
            nodes = [n for n in have if n != own] or have or [0]
            if len(have) <= 2:
                nodes = have or [0]
    set_cxl_nodes(nodes)
    return nodes


def load_tokenizer(model_id):
    """The tokenizer a local checkpoint directory ships (run_generation.py:167, `from_pretrained(args.model_id)`), or None: a shape
    name (no files offline) or a directory without tokenizer files."""
    if not os.path.isdir(model_id) or not any(os.path.exists(os.path.join(model_id, f)) for f in TOKENIZER_FILES):
        return None
    from transformers import AutoTokenizer
    return AutoTokenizer.from_pretrained(model_id, local_files_only=True)


def prompt_input_ids(args, vocab, tokenizer, out=print):
    """run_generation.py:262-285: `--prompt` goes through the checkpoint's tokenizer, every row of the batch is that prompt
    (`prompt = [prompt] * args.batch_size`), and its size is printed.  The reference otherwise draws the text from its prompt.json
    pool by --input-tokens; that file's texts are not reproduced here, so without --prompt the ids are the synthetic row of
    --input-tokens tokens as before."""
    if args.prompt is None:
        return synthetic_prompt(vocab, int(args.input_tokens), args.batch_size)
    if tokenizer is None:
        raise SystemExit("[ERROR] --prompt needs a tokenizer: point -m at a checkpoint directory that ships its tokenizer files "
                         f"({', '.join(TOKENIZER_FILES[:3])}); {args.model_id!r} has none")
    row = tokenizer(args.prompt, return_tensors="pt").input_ids.to(torch.int64)
    if row.numel() == 0 or int(row.max()) >= vocab or int(row.min()) < 0:
        raise SystemExit(f"[ERROR] the tokenizer produced ids outside the model's vocabulary of {vocab} entries")
    out("---- Prompt size:", int(row.shape[1]))
    return row.repeat(args.batch_size, 1)


def is_llama(args):
    """`-m` names a Llama: a HF directory whose config.json says so (run_generation.py:159-166 loads OPT and Llama through the same
    AutoModelForCausalLM call), or one of the known Llama shape names (random-init weights, like the OPT shape names)"""
    if os.path.isdir(args.model_id):
        from .checkpoint import is_llama_dir
        return is_llama_dir(args.model_id)
    try:
        resolve_shape(args.model_id)
        return False
    except ValueError:
        from .llama import resolve_llama_shape
        try:
            resolve_llama_shape(args.model_id)
            return True
        except ValueError:
            return False


def model_shape(args):
    """shape of the model `-m` names, without loading it (the planner needs it first)"""
    if is_llama(args):
        from .checkpoint import llama_shape_of
        from .llama import resolve_llama_shape
        return llama_shape_of(args.model_id) if os.path.isdir(args.model_id) else resolve_llama_shape(args.model_id)
    if os.path.isdir(args.model_id):
        import json
        from . import packed_checkpoint
        from .model import OPTShape
        if packed_checkpoint.is_packed_dir(args.model_id):
            sh = json.load(open(os.path.join(args.model_id, packed_checkpoint.MANIFEST)))["shape"]
            return OPTShape(sh["name"], sh["hidden"], sh["heads"], sh["ffn"], sh["layers"], vocab=sh["vocab"], max_pos=sh["max_pos"])
        cfg = json.load(open(os.path.join(args.model_id, "config.json")))
        return OPTShape(os.path.basename(args.model_id.rstrip("/")), cfg["hidden_size"], cfg["num_attention_heads"], cfg["ffn_dim"],
                        cfg["num_hidden_layers"], vocab=cfg["vocab_size"], max_pos=cfg["max_position_embeddings"])
    return resolve_shape(args.model_id)


def auto_plan(args, out=print):
    """--auto-plan: calibrate the box, plan, overwrite the LIA flags with the plan (f-3)."""
    from . import planner
    box = planner.calibrate(verbose=False)
    if args.plan_hbm_gb:                           # what-if: pretend the GPU is smaller
        box.hbm_gb = float(args.plan_hbm_gb)
    max_pct = int(args.plan_max_gpu_percentage)    # what-if / tests: cap the resident share
    fmt = args.stream_format or default_stream_format()
    box.wire_ratio = {"raw": 1.0, "pack10": 0.675}[fmt]
    shape = model_shape(args)
    pl = planner.plan(shape, args.batch_size, int(args.input_tokens), args.max_new_tokens, box, max_gpu_percentage=max_pct)
    args.gpu_percentage, args.prefill_policy, args.decoding_policy = pl.gpu_percentage, pl.prefill_policy, pl.decoding_policy
    args.num_minibatch, args.pin_weight, args.stream_format = pl.num_minibatch, True, fmt
    coop_note = ""
    if pl.n_gpu_layers < shape.layers and pl.decoding_policy in (2, 3):
        # layers stream: the host cores take some of them (build-defined cooperative split).  The plan SEEDS the count; the
        # scheduler's online controller (cpu_layers = -1) moves it with the measured decode steps.  With room in HBM for every
        # layer's KV cache the other streamed layers attend on the GPU (policies 3/3): the host cores then run nothing but their
        # own layers (OPT-30B gpu% = 10 on MI355X: 190-205 against 165-185 tokens/s)
        B_, T_, new_ = args.batch_size, int(args.input_tokens), args.max_new_tokens
        c2, ms2 = planner.plan_cpu_layers(shape, B_, T_, new_, pl.gpu_percentage, box)
        c3, ms3 = planner.plan_cpu_layers(shape, B_, T_, new_, pl.gpu_percentage, box, kv_in_hbm=True)
        hbm3 = planner.estimate(shape, B_, T_, new_, pl.gpu_percentage, 3, box)[2]
        if c3 > 0 and ms3 < ms2 and hbm3 <= 0.92 * box.hbm_gb:
            args.prefill_policy, args.decoding_policy, args.cpu_layers_start, ms = 3, 3, c3, ms3
        else:
            args.prefill_policy, args.decoding_policy, args.cpu_layers_start, ms = 0, 2, c2, ms2
        args.cpu_layers = -1 if args.cpu_layers_start > 0 else 0
        if args.cpu_layers:
            coop_note = f"; cooperative split predicted {1e3 * B_ / ms:.1f} tokens/s at {args.cpu_layers_start} host layers"
    out(f"auto-plan: calibrated {box.calibrated}")
    out(f"auto-plan: gpu%={pl.gpu_percentage} ({pl.n_gpu_layers} resident layers) prefill policy {args.prefill_policy} decode policy "
        f"{args.decoding_policy} cpu-layers {'online from ' + str(args.cpu_layers_start) if args.cpu_layers < 0 else args.cpu_layers} wire {fmt}; predicted prefill {pl.prefill_ms:.0f} ms, "
        f"{pl.decode_tokens_per_s:.1f} tokens/s ({pl.note}){coop_note}")
    return pl


def load_model(args):
    if is_llama(args):
        # build-defined (the reference's LlamaDecoderLayer_forward carries no policy, decoder.py:121-169): layers [0, gpu%) resident,
        # the others stream through the same WeightPipeline, every layer's arithmetic and KV cache on the GPU
        from .llama import LiaLlamaModel
        shape = model_shape(args)
        n_gpu = shape.layers if args.gpu_percentage >= 100 else int(shape.layers * args.gpu_percentage / 100)
        fmt = {"raw": 0, "pack10": 10}[args.stream_format or default_stream_format()]
        if os.path.isdir(args.model_id):
            from .checkpoint import load_hf_llama
            return load_hf_llama(args.model_id, n_gpu_layers=n_gpu, pin_weight=True, wire=fmt)
        return LiaLlamaModel.random_init(shape, seed=args.seed, n_gpu_layers=n_gpu, pin_weight=True, pack=fmt)
    if os.path.isdir(args.model_id):
        from . import packed_checkpoint
        if packed_checkpoint.is_packed_dir(args.model_id):
            import json
            man = json.load(open(os.path.join(args.model_id, packed_checkpoint.MANIFEST)))
            shape_layers = man["shape"]["layers"]
            if args.stream_format is None:       # stream what is on disk, as it is
                wires = [e["wire"] for e in man["layers"]]
                args.stream_format = {0: "raw", 10: "pack10"}.get(max(set(wires), key=wires.count), "pack10")     # (load_packed refuses other formats by name)
            return packed_checkpoint.load_packed(args.model_id, n_gpu_layers=int(shape_layers * args.gpu_percentage / 100))
        # a HF directory: converted layer by layer, every layer straight to the tier the flags name (checkpoint.load_hf_opt)
        from .checkpoint import load_hf_opt
        from .scheduler import OffloadScheduler
        import json
        L_ = json.load(open(os.path.join(args.model_id, "config.json")))["num_hidden_layers"]
        n_gpu = int(L_ * args.gpu_percentage / 100)
        fmt = {"raw": 0, "pack10": 10}[args.stream_format or default_stream_format()]
        raw = OffloadScheduler.cpu_layer_set(n_gpu, L_, args.cpu_layers) if (args.cpu_layers and args.cpu_layers > 0 and args.decoding_policy in (2, 3)) else ()
        from .scheduler import placement_formats
        fmt, raw = placement_formats(args.prefill_policy, args.decoding_policy, fmt, n_gpu, L_, args.pin_weight, args.enable_cxl, raw)
        return load_hf_opt(args.model_id, n_gpu_layers=n_gpu, pin_weight=args.pin_weight, enable_cxl=args.enable_cxl, wire=fmt, raw_layers=raw)
    shape = resolve_shape(args.model_id)
    n_gpu = int(shape.layers * args.gpu_percentage / 100)
    fmt = {"raw": 0, "pack10": 10}[args.stream_format or default_stream_format()]
    from .scheduler import OffloadScheduler, placement_formats
    raw = OffloadScheduler.cpu_layer_set(n_gpu, shape.layers, args.cpu_layers) if (args.cpu_layers and args.cpu_layers > 0 and args.decoding_policy in (2, 3)) else ()
    # the host path reads a raw copy in place; a 0 / 1 line keeps the packed copy for the prefill's stream beside it (placement_formats)
    fmt, raw = placement_formats(args.prefill_policy, args.decoding_policy, fmt, n_gpu, shape.layers, args.pin_weight, args.enable_cxl, raw)
    return LiaOPTModel.random_init(shape, seed=args.seed, init=args.init, n_gpu_layers=n_gpu,
                                   pin_weight=args.pin_weight, enable_cxl=args.enable_cxl, wire=fmt, raw_layers=raw)


def summarize(total_time, num_iter, num_warmup, total_list, batch_size, out=print):
    """The reference's summary block (run_generation.py:337-354), verbatim formats; returns the numbers."""
    out("\n", "-" * 10, "Summary:", "-" * 10)
    latency = total_time / (num_iter - num_warmup)
    out("Inference latency: %.3f sec." % latency)
    res = {"inference_latency_s": latency}
    if total_list:
        from itertools import chain
        first_latency = float(np.mean([x[0] for x in total_list]))
        average_2n = sorted(chain(*[x[1:] for x in total_list]))
        average_2n_latency = float(np.mean(average_2n))
        p90_latency = average_2n[int(len(average_2n) * 0.9)]
        p99_latency = average_2n[int(len(average_2n) * 0.99)]
        out("First token average latency: %.3f sec." % first_latency)
        out("Average 2... latency: %.3f sec." % average_2n_latency)
        out("P90 2... latency: %.3f sec." % p90_latency)
        out("P99 2... latency: %.3f sec." % p99_latency)
        res.update(prefill_ms=1e3 * first_latency, decode_tokens_per_s=batch_size / average_2n_latency,
                   p90_s=p90_latency, p99_s=p99_latency)
    return res


def profile_once(model, input_ids, generate_kwargs, out=print, warm=True):
    """--profile (run_generation.py:290-307): one generate() with every GEMM launch bracketed by HIP events on its own stream
    (lia_prof_*), the host-attention wall clock and the weight stream's copy-engine time; prints the table and returns it.
    warm=False (--result-json): no untimed call in front -- the profiled generation IS the harness's first warm-up iteration."""
    sched = model._lia_scheduler
    if warm:
        generate(model, input_ids, **dict(generate_kwargs, max_steps=2))       # untimed: allocations, placement, page-in
    elif getattr(sched, "ctx", None) is None:
        generate(model, input_ids, **dict(generate_kwargs, max_steps=1))       # the library context (and its brackets) exists after the first forward
    st = {}

    def host_time(reset=True):
        ht = dict(getattr(sched, "host_layer_time", None) or {"ms": 0.0, "calls": 0})
        if reset and hasattr(sched, "host_layer_time"):
            sched.host_layer_time = {"ms": 0.0, "calls": 0}
        return ht

    def hook(step):
        if step == 0:
            sched.stream_stats(reset=True)
            host_time()
            sched.ctx.prof_start(65536)
        elif step == 1:
            st["prefill"], st["prefill_h2d"], st["prefill_host"] = sched.ctx.prof_stop(), sched.stream_stats(reset=True), host_time()
            sched.ctx.prof_start(65536)

    res = generate(model, input_ids, step_hook=hook, **dict(generate_kwargs, token_latency=True))
    lat = res[1]
    st["decode"], st["decode_h2d"], st["decode_host"] = sched.ctx.prof_stop(), sched.stream_stats(), host_time()
    if "prefill" not in st:          # a one-token generation: everything is the prefill
        st["prefill"], st["prefill_h2d"], st["decode"], st["decode_h2d"] = st["decode"], st["decode_h2d"], None, (0.0, 0.0)
        st["prefill_host"], st["decode_host"] = st["decode_host"], {"ms": 0.0, "calls": 0}
    rows = []
    for phase, wall in (("prefill", lat[0]), ("decode", sum(lat[1:]))):
        pr = st[phase]
        if pr is None:
            continue
        b, ms = st[phase + "_h2d"]
        for regime in ("tiled", "skinny"):
            n = pr[regime + "_launches"]
            if n:
                t = pr[regime + "_ms"]
                rows.append((phase, f"GEMM {regime} (M {'> 256' if regime == 'tiled' else '<= 256'})", n, t, f"{pr[regime + '_flops'] / t / 1e9:.1f} TFLOP/s",
                             f"{pr[regime + '_bytes'] / t / 1e6:.0f} GB/s"))
        if pr["host_attention_calls"]:
            rows.append((phase, "host attention (policy 2)", pr["host_attention_calls"], pr["host_attention_ms"], "", ""))
        hl = st.get(phase + "_host") or {}
        if hl.get("calls"):
            rows.append((phase, "host layers (policy 1: linears + attention on the host cores)", hl["calls"], hl["ms"], "", ""))
        if b:
            rows.append((phase, "weight stream H2D (copy engine busy)", "", ms, "", f"{b / max(ms, 1e-9) / 1e6:.1f} GB/s"))
        rows.append((phase, "wall clock", "", 1e3 * wall, "", ""))
    out("\n" + "-" * 10 + " Profile (one generate; HIP-event brackets per launch) " + "-" * 10)
    out("%-8s %-40s %8s %12s %16s %12s" % ("phase", "what", "calls", "ms", "compute", "memory"))
    for r in rows:
        out("%-8s %-40s %8s %12.3f %16s %12s" % r)
    profile_once.last = (res, rows)
    return rows


def _dominant(rows, phase):
    """the largest time share of a phase among the profiled resources (GEMM regimes, host attention, the link's copy engine)"""
    cand = [(r[3], r[1]) for r in rows if r[0] == phase and r[1] != "wall clock"]
    wall = next((r[3] for r in rows if r[0] == phase and r[1] == "wall clock"), None)
    if not cand or not wall:
        return None
    ms, what = max(cand)
    return {"what": what, "ms": round(ms, 2), "share_of_wall": round(ms / wall, 3)}


def plan_beside(args):
    """what lia_amd.planner would pick for this line's model / batch / lengths on this box (not applied): the reference hand-picks
    these per script line (llm/scripts/lia_offline.sh:13-29)"""
    if is_llama(args):
        return {"note": "lia_amd.planner models OPT layers and the LIA policies; a Llama runs every layer on the GPU (build-defined, SURVEY.md quirk 3)"}
    try:
        from . import planner
        box = planner.calibrate(verbose=False)
        fmt = args.stream_format or default_stream_format()
        box.wire_ratio = {"raw": 1.0, "pack10": 0.675}[fmt]
        shape = model_shape(args)
        B, T, new = args.batch_size, int(args.input_tokens), args.max_new_tokens
        pl = planner.plan(shape, B, T, new, box)
        rec = {"gpu_percentage": pl.gpu_percentage, "prefill_policy": pl.prefill_policy, "decoding_policy": pl.decoding_policy,
               "predicted_prefill_ms": round(pl.prefill_ms, 1), "predicted_decode_tokens_per_s": round(pl.decode_tokens_per_s, 1),
               "hbm_gb": round(pl.hbm_gb, 1), "host_gb": round(pl.host_gb, 1), "note": pl.note, "calibrated": box.calibrated}
        if pl.n_gpu_layers < shape.layers:
            c2, ms2 = planner.plan_cpu_layers(shape, B, T, new, pl.gpu_percentage, box)
            c3, ms3 = planner.plan_cpu_layers(shape, B, T, new, pl.gpu_percentage, box, kv_in_hbm=True)
            rec["cooperative"] = {"kv_on_host": {"host_layers": c2, "predicted_tokens_per_s": round(1e3 * B / ms2, 1)},
                                  "kv_in_hbm": {"host_layers": c3, "predicted_tokens_per_s": round(1e3 * B / ms3, 1)}}
        try:          # the hand-picked flags through the same model, for the ratio
            pre, dec, hbm, host, _ = planner.estimate(shape, B, T, new, args.gpu_percentage, args.decoding_policy if args.decoding_policy in (2, 3) else 2, box)
            rec["hand_picked_estimate"] = {"prefill_ms": round(pre, 1), "decode_tokens_per_s": round(1e3 * B / dec, 1), "hbm_gb": round(hbm, 1),
                                           "host_gb": round(host, 1), "note": "decode estimated as policy 2" if args.decoding_policy not in (2, 3) else ""}
        except Exception as e:       # noqa: BLE001
            rec["hand_picked_estimate"] = {"error": str(e)}
        return rec
    except MemoryError as e:
        return {"error": f"no placement fits: {e}"}


def main(argv=None):
    args = build_parser().parse_args(argv)
    if not args.result_json:
        return _main(args)
    import json
    import resource
    from . import hostinfo
    rec = {"argv": list(argv) if argv is not None else None, "flags": {k: v for k, v in vars(args).items() if k != "result_json"}}
    t0 = time.time()
    mem0 = hostinfo.cgroup_memory()["current"]
    rec["_mem_samples"] = []
    try:
        rec["result"] = _main(args, rec)
        rec["status"] = "ok"
    except MemoryError as e:
        rec["status"], rec["reason"] = "refused: memory", str(e)
        print("[refused]", e)
    except ValueError as e:
        rec["status"], rec["reason"] = "refused: flags", str(e)
        print("[refused]", e)
    rec["wall_s"] = round(time.time() - t0, 1)
    mem = hostinfo.cgroup_memory()
    samples = [v for v in rec.pop("_mem_samples", []) + [mem["current"]] if v is not None]
    rec["host_memory"] = {"container_gib_at_start": None if mem0 is None else round(mem0 / 2**30, 2),
                          "container_gib_peak_sampled": round(max(samples) / 2**30, 2) if samples else None,
                          "this_run_gib": round((max(samples) - mem0) / 2**30, 2) if (samples and mem0 is not None) else None,
                          "sampled": "memory.current of the container after the model load and after every iteration (pinned weights + host KV caches + page cache)",
                          "cgroup_limit_gib": None if mem["max"] is None else round(mem["max"] / 2**30, 1),
                          "process_max_rss_gib": round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 2**20, 2)}
    os.makedirs(os.path.dirname(os.path.abspath(args.result_json)), exist_ok=True)
    with open(args.result_json, "w") as f:
        json.dump(rec, f, indent=1, default=str)
    return rec.get("result")


def _main(args, rec=None):
    print(args)
    if not args.benchmark:
        print("note: only the --benchmark protocol exists here; running it")
    # the reference's launch line is `OMP_NUM_THREADS=40 numactl -m 0 -C 0-39 python run.py ...` (README.md:78): host compute
    # on the node that holds the pinned weights / KV caches.  Here: the NUMA node of GPU 0 (LIA_PIN_NODE overrides, -1 = off).
    import os
    from . import hostinfo
    import torch
    if torch.cuda.is_available():
        node = hostinfo.pin_node(0)
        if node >= 0 and hostinfo.pin_to_node(node):
            print(f"host threads pinned to NUMA node {node}")
    if args.enable_cxl:
        nodes = choose_cxl_nodes(args.cxl_nodes)
        if nodes is not None:
            print(f"--enable-cxl: the NUMA tier interleaves over nodes {nodes}")
            if rec is not None:
                rec["cxl_nodes"] = nodes
    if rec is not None and not args.auto_plan:
        rec["planner_pick"] = plan_beside(args)
    if args.auto_plan and is_llama(args):
        raise SystemExit("[ERROR] --auto-plan plans OPT layers and the LIA policies; a Llama has no policy to plan (pick --gpu-percentage)")
    if args.auto_plan:
        pl = auto_plan(args)
        if rec is not None:
            rec["auto_plan"] = {"gpu_percentage": args.gpu_percentage, "prefill_policy": args.prefill_policy, "decoding_policy": args.decoding_policy,
                                "cpu_layers": args.cpu_layers, "cpu_layers_start": args.cpu_layers_start,
                                "predicted_prefill_ms": round(pl.prefill_ms, 1), "predicted_decode_tokens_per_s": round(pl.decode_tokens_per_s, 1)}
    t_load = time.time()
    model = load_model(args)
    if rec is not None:
        rec["model_load_s"] = round(time.time() - t_load, 1)
        from . import hostinfo as _hi
        rec["_mem_samples"].append(_hi.cgroup_memory()["current"])
    if args.stream_format is None:
        args.stream_format = default_stream_format()
    from .scheduler import OffloadScheduler
    if getattr(model, "family", "opt") == "llama":
        from .llama import LlamaScheduler
        model._lia_scheduler = LlamaScheduler(model, pack=args.stream_format)
        print(f"note: {model.shape.name} is a Llama: --prefill-policy / --decoding-policy have no counterpart (decoder.py:121-169); "
              f"gpu% {args.gpu_percentage} of the layers resident, the others streamed, every layer computed on the GPU")
    else:
        model._lia_scheduler = OffloadScheduler(model, wire=args.stream_format)     # generate() drives this scheduler
    generate_kwargs = dict(do_sample=False, num_beams=1, max_new_tokens=args.max_new_tokens, min_new_tokens=args.max_new_tokens,
                           token_latency=args.token_latency, prefill_policy=args.prefill_policy,
                           decoding_policy=args.decoding_policy, no_overlap=args.no_overlap, pin_weight=args.pin_weight,
                           gpu_percentage=args.gpu_percentage, num_minibatch=args.num_minibatch, enable_cxl=args.enable_cxl)
    if args.cpu_layers:
        generate_kwargs["cpu_layers"] = args.cpu_layers
        if args.cpu_layers < 0 and args.cpu_layers_start is not None:
            generate_kwargs["cpu_layers_start"] = args.cpu_layers_start
    tokenizer = load_tokenizer(args.model_id)
    input_ids = prompt_input_ids(args, model.shape.vocab, tokenizer)
    if args.profile:
        profile_once(model, input_ids, generate_kwargs)
    total_time, total_list = 0.0, []
    sched = model._lia_scheduler
    for i in range(args.num_iter):
        if rec is not None and i == args.num_warmup and hasattr(sched, "stream_stats"):
            sched.stream_stats(reset=True)                     # the weight stream's bytes / copy-engine time over the timed iterations
        tic = time.time()
        if rec is not None and i == 0 and args.num_warmup >= 1 and hasattr(sched, "coop_report"):
            profile_once(model, input_ids, generate_kwargs, warm=False)       # the first warm-up iteration, bracketed
            output = profile_once.last[0]
            if not args.token_latency:
                output = output[0]
            rows = profile_once.last[1]
            rec["profile_of_warmup_iteration"] = {"rows": [list(r) for r in rows], "dominant_prefill": _dominant(rows, "prefill"),
                                                  "dominant_decode": _dominant(rows, "decode")}
        else:
            output = generate(model, input_ids, **generate_kwargs)
        gen_ids = output[0] if args.token_latency else output
        toc = time.time()
        total_new_tokens = [int(o.shape[0] - i_.shape[0]) for i_, o in zip(input_ids, gen_ids)]
        if tokenizer is not None and args.prompt is not None:                 # run_generation.py:321-323
            print(tokenizer.batch_decode(gen_ids, skip_special_tokens=True)[:1], total_new_tokens[:4], flush=True)
        print(gen_ids[0, input_ids.shape[1]:].tolist(), total_new_tokens[:4], flush=True)
        print("Iteration: %d, Time: %.6f sec" % (i, toc - tic), flush=True)
        if rec is not None:
            from . import hostinfo as _hi
            rec["_mem_samples"].append(_hi.cgroup_memory()["current"])
        if i >= args.num_warmup:
            total_time += toc - tic
            if args.token_latency:
                total_list.append(output[1])
    res = summarize(total_time, args.num_iter, args.num_warmup, total_list, args.batch_size)
    if rec is not None and hasattr(sched, "stream_stats"):
        b, ms = sched.stream_stats()
        n_t = max(1, args.num_iter - args.num_warmup)
        rec["weight_stream"] = {"bytes_per_timed_iteration": b / n_t, "copy_engine_busy_ms_per_timed_iteration": round(ms / n_t, 1),
                                "gbs_while_busy": round(b / max(ms, 1e-9) / 1e6, 2), "gbs_over_wall": round(b / max(total_time, 1e-9) / 1e9, 2),
                                "fraction_of_63_gbs_link_over_wall": round(b / max(total_time, 1e-9) / 63e9, 3),
                                "wire_format": ("pack10" if any(getattr(st, "packed", 0) for st in model.layers) else "raw"),
                                "wire_format_note": "the format of the host copies that streamed (a policy-1 / NUMA-tier / unpinned placement ships raw: scheduler.placement_formats)"}
    if "decode_tokens_per_s" in res:
        print("Decode throughput: %.2f tokens/s, prefill %.1f ms" % (res["decode_tokens_per_s"], res["prefill_ms"]))
    coop = getattr(model._lia_scheduler, "coop_report", lambda: None)()
    if coop:
        print("Cooperative split (online): %d host-computed decode layers, ms per step by count %s" % (coop["host_layers"], coop["ms_by_count"]))
    return res


if __name__ == "__main__":
    main()
