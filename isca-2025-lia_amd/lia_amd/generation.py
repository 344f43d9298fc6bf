"""generate() + the greedy loop: counterpart of GenerationMixin.generate with the LIA edits
(lia/generation_utils.py:1166-1170, 1542-1558) and IPEX's _greedy_search
(intel_extension_for_pytorch/transformers/generation/greedy_search.py:37-458).

Same call convention as the reference harness uses (llm/single_instance/run_generation.py:179-182,
319-320):

    out = generate(model, input_ids, do_sample=False, num_beams=1, max_new_tokens=N, min_new_tokens=N,
                   prefill_policy=, decoding_policy=, no_overlap=, pin_weight=, gpu_percentage=,
                   num_minibatch=, enable_cxl=, token_latency=True)
    ids, latency_list = out        # (ids only when token_latency is False)

latency_list[i] is the wall clock of greedy iteration i (prepare inputs -> forward -> lm_head -> argmax
-> append), exactly the quantity greedy_search.py:145,424 records: [0] is the prefill, [1:] the decode steps.
"""
import time

import torch

from .scheduler import KVState, OffloadScheduler

LIA_KWARGS = ("prefill_policy", "decoding_policy", "no_overlap", "pin_weight", "gpu_percentage", "num_minibatch",
              "enable_cxl")


def _scheduler_of(model):
    if getattr(model, "_lia_scheduler", None) is None:
        if getattr(model, "family", "opt") == "llama":
            from .llama import LlamaScheduler
            model._lia_scheduler = LlamaScheduler(model)
        else:
            model._lia_scheduler = OffloadScheduler(model)
    return model._lia_scheduler


def generate(model, input_ids, max_new_tokens=None, min_new_tokens=None, do_sample=False, num_beams=1,
             eos_token_id=2, pad_token_id=1, token_latency=False, return_logits=False, **model_kwargs):
    if do_sample or num_beams != 1:
        raise ValueError("only greedy search carries the LIA kwargs (run.py --greedy; beam/sample paths are not plumbed "
                         "in the reference either)")
    if max_new_tokens is None or max_new_tokens < 1:
        raise ValueError("max_new_tokens must be >= 1")
    # unknown kwargs are NOT rejected: the reference disables that check so its flags reach forward()
    # (lia/generation_utils.py:1166-1170)
    lia = {k: model_kwargs.pop(k, None) for k in LIA_KWARGS}
    lia = {"prefill_policy": 1 if lia["prefill_policy"] is None else lia["prefill_policy"],
           "decoding_policy": 1 if lia["decoding_policy"] is None else lia["decoding_policy"],
           "no_overlap": bool(lia["no_overlap"]), "pin_weight": bool(lia["pin_weight"]),
           "gpu_percentage": lia["gpu_percentage"] or 0, "num_minibatch": lia["num_minibatch"] or 1,
           "enable_cxl": bool(lia["enable_cxl"])}
    if model_kwargs.get("cpu_layers"):           # build-defined extension (scheduler.forward): host-computed decode layers
        lia["cpu_layers"] = int(model_kwargs.pop("cpu_layers"))          # -1: chosen online (scheduler.CoopController)
        if model_kwargs.get("cpu_layers_start") is not None:
            lia["cpu_layers_start"] = int(model_kwargs.pop("cpu_layers_start"))
    model_kwargs.pop("cpu_layers_start", None)
    # build-defined, for bench.py: step_hook(i) runs before greedy iteration i (barriers / profiler brackets around exactly the
    # timed steps); max_steps ends the loop early while the caches stay sized for max_new_tokens (a short warm-up call)
    hooks = {"step_hook": model_kwargs.pop("step_hook", None), "max_steps": model_kwargs.pop("max_steps", None)}
    min_new = min_new_tokens or 0
    return _greedy_search(model, input_ids, max_new_tokens, min_new, eos_token_id, pad_token_id, token_latency,
                          return_logits, lia, **hooks)


def _greedy_search(model, input_ids, max_new_tokens, min_new_tokens, eos_token_id, pad_token_id, token_latency,
                   return_logits, lia, step_hook=None, max_steps=None):
    sched = _scheduler_of(model)
    ids = torch.as_tensor(input_ids, dtype=torch.int64).cpu()
    if ids.dim() != 2:
        raise ValueError("input_ids must be [batch, seq]")
    B, T = ids.shape
    if T + max_new_tokens > model.shape.max_pos:
        raise ValueError(f"prompt {T} + max_new_tokens {max_new_tokens} exceeds max positions {model.shape.max_pos}")
    L = model.shape.layers
    n_gpu = int(L * lia["gpu_percentage"] / 100)
    if getattr(model, "family", "opt") == "llama":
        from .llama import LlamaKVState
        kv = LlamaKVState(model, B, T + max_new_tokens)
    else:
        # caches sized [T+new, B, h, d] like modeling_opt.py:1277-1278; in HBM for every layer when both policies are 3
        on_dev = lia["prefill_policy"] == 3 and lia["decoding_policy"] == 3
        host_layers, dual_layers = (), ()
        if on_dev and lia.get("cpu_layers", 0) > 0:
            host_layers = OffloadScheduler.cpu_layer_set(n_gpu, L, lia["cpu_layers"])
        elif on_dev and lia.get("cpu_layers", 0) < 0 and getattr(sched, "dp", None) is None and n_gpu < L - 1:
            # online count: every CANDIDATE host layer has a cache buffer on both sides; the cache lives where the layer is computed
            coop = sched._coop_controller(n_gpu, L, B, T, max_new_tokens, lia["gpu_percentage"], 3, lia.get("cpu_layers_start"))
            # (before the caches are pinned: a candidate costs its raw weight copy AND its host cache buffer; while copies are
            # missing, the pooled blocks of an earlier generation with the caches on the host go back first -- with them in HBM
            # only the candidates need one)
            sched._fit_host_candidates(coop, bool(lia.get("enable_cxl") and lia.get("pin_weight")),
                                       extra_per_layer=2 * (T + max_new_tokens) * B * model.shape.hidden * 2, trim_pool=True)
            host_layers, dual_layers = coop.host_set(), coop.superset()
        kv = KVState(model, n_gpu, B, T + max_new_tokens, all_on_device=on_dev, host_layers=host_layers, dual_layers=dual_layers)
    unfinished = torch.ones(B, dtype=torch.int64)
    all_unfinished = True
    latency_list, logits_list = [], []
    # the ids grow in place (greedy_search.py:408 re-allocates [B, T + t] with torch.cat every step: 1 MB per step at
    # B 128 x T 1024, on the critical path between two decode steps)
    ids_buf = torch.empty((B, T + max_new_tokens), dtype=torch.int64)
    ids_buf[:, :T] = ids
    cur = ids
    step = 0
    while True:
        if step_hook is not None:
            step_hook(step)
        tic = time.time()
        # EOS is suppressed while fewer than min_new_tokens were generated (HF MinNewTokensLengthLogitsProcessor)
        suppress = eos_token_id if (eos_token_id is not None and step < min_new_tokens) else -1
        logits, nxt = sched.forward(cur, kv, max_new_tokens=max_new_tokens, suppress_token=suppress, **lia)
        next_tokens = nxt.cpu()
        same_as_device = eos_token_id is None or all_unfinished        # no row was replaced by the pad token below
        if eos_token_id is not None:
            next_tokens = next_tokens * unfinished + pad_token_id * (1 - unfinished)   # greedy_search.py:398-405
        ids_buf[:, T + step] = next_tokens                                             # :408
        if eos_token_id is not None:
            unfinished = unfinished * (next_tokens != eos_token_id).long()             # :415-421
            all_unfinished = bool(unfinished.min() == 1)
        # the next step's input: the argmax output where it already sits in HBM, unless the EOS bookkeeping changed a row
        cur = nxt.view(B, 1) if (same_as_device and nxt.is_cuda) else next_tokens[:, None]
        step += 1
        if return_logits:
            logits_list.append(logits.clone())
        latency_list.append(time.time() - tic)                                         # :424
        if (eos_token_id is not None and not all_unfinished and unfinished.max() == 0) or step >= max_new_tokens or \
                (max_steps is not None and step >= max_steps):
            break
    out = ids_buf[:, :T + step].clone() if step < max_new_tokens else ids_buf
    if return_logits:
        return out, latency_list, logits_list
    if token_latency:
        return out, latency_list
    return out
