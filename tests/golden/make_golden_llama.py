#!/usr/bin/env python3
"""Golden vectors for the Llama-family layer (BASELINE.json config 4).  The reference tree has no LIA Llama path
(LlamaDecoderLayer_forward takes no policy, decoder.py:121-169), so the arithmetic of record is
stock HF transformers' eager Llama in bf16, executed here on CPU: per-layer outputs (prefill + decode through
LlamaDecoderLayer with a DynamicCache) and end-to-end greedy ids.  Inputs are re-derived from seeds (synth.py)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import synth  # noqa: E402
from make_golden import bits_to_torch, torch_to_bits  # noqa: E402


def hf_model(m, vocab, H, heads, kv_heads, F, L, max_pos, theta, dtype):
    from transformers import LlamaConfig, LlamaForCausalLM
    cfg = LlamaConfig(vocab_size=vocab, hidden_size=H, intermediate_size=F, num_hidden_layers=L, num_attention_heads=heads,
                      num_key_value_heads=kv_heads, max_position_embeddings=max_pos, rms_norm_eps=1e-5, rope_theta=theta,
                      tie_word_embeddings=False, attention_bias=False, mlp_bias=False, hidden_act="silu", bos_token_id=1,
                      eos_token_id=2, pad_token_id=0, attn_implementation="eager")
    model = LlamaForCausalLM(cfg).to(dtype).eval()
    # .to(bf16) also rounds the rotary inv_freq buffer to bf16; from_pretrained(torch_dtype=bf16) -- what a user of the
    # real checkpoint gets -- keeps that non-persistent buffer in fp32.  Restore the fp32 values.
    d = H // heads
    model.model.rotary_emb.inv_freq = 1.0 / (theta ** (torch.arange(0, d, 2, dtype=torch.int64).float() / d))
    sd = {"model.embed_tokens.weight": bits_to_torch(m["embed_tokens"]), "lm_head.weight": bits_to_torch(m["lm_head"]),
          "model.norm.weight": bits_to_torch(m["final_norm_w"])}
    names = {"in_norm_w": "input_layernorm.weight", "q_w": "self_attn.q_proj.weight", "k_w": "self_attn.k_proj.weight",
             "v_w": "self_attn.v_proj.weight", "o_w": "self_attn.o_proj.weight", "post_norm_w": "post_attention_layernorm.weight",
             "gate_w": "mlp.gate_proj.weight", "up_w": "mlp.up_proj.weight", "down_w": "mlp.down_proj.weight"}
    for i, lw in enumerate(m["layers"]):
        for n, v in lw.items():
            sd[f"model.layers.{i}.{names[n]}"] = bits_to_torch(v)
    res = model.load_state_dict({k: v.to(dtype) for k, v in sd.items()}, strict=False)
    assert not res.unexpected_keys and all("rotary" in k or "inv_freq" in k for k in res.missing_keys), res
    return model


def run_layer_case(name, H, heads, kv_heads, F, B, T, new, seed, w_std, theta=10000.0):
    """One-layer model: hidden states after layer 0 for a prefill and `new` decode steps (inputs_embeds path)."""
    m = synth.make_llama_model(seed, 64, H, heads, kv_heads, F, 1, w_std)
    model = hf_model(m, 64, H, heads, kv_heads, F, 1, T + new + 4, theta, torch.bfloat16)
    x = bits_to_torch(synth.make_hidden(seed + 1, B, T, H))
    out = {"cfg": np.array([H, heads, kv_heads, F, B, T, new, seed], dtype=np.int64), "w_std": np.array([w_std]),
           "theta": np.array([theta])}
    with torch.no_grad():
        o = model.model(inputs_embeds=x, use_cache=True, output_hidden_states=True)
        out["prefill_hidden"] = torch_to_bits(o.hidden_states[1])          # layer 0 output AFTER the model's final RMSNorm
        past = o.past_key_values
        for s in range(new):
            xs = bits_to_torch(synth.make_hidden(seed + 100 + s, B, 1, H))
            o = model.model(inputs_embeds=xs, past_key_values=past, use_cache=True, output_hidden_states=True)
            out[f"dec{s}_hidden"] = torch_to_bits(o.hidden_states[1])
            past = o.past_key_values
        k, v = past.layers[0].keys, past.layers[0].values                  # [B, kvh, S, d], post-RoPE keys
        out["kcache"] = torch_to_bits(k.permute(2, 0, 1, 3))               # -> seq-major [S, B, kvh, d]
        out["vcache"] = torch_to_bits(v.permute(2, 0, 1, 3))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, {k_: v_.shape for k_, v_ in out.items() if k_ not in ("cfg", "w_std", "theta")})


def run_generate_case(name, vocab, H, heads, kv_heads, F, L, B, T, new, seed0, w_std, theta=10000.0, min_gap=0.12):
    for seed in range(seed0, seed0 + 200):
        m = synth.make_llama_model(seed, vocab, H, heads, kv_heads, F, L, w_std)
        ids = torch.from_numpy(synth.make_prompt_ids(seed + 1, B, T, vocab))
        res = {"cfg": np.array([vocab, H, heads, kv_heads, F, L, B, T, new, seed], dtype=np.int64), "w_std": np.array([w_std]),
               "theta": np.array([theta])}
        for dt, tag in ((torch.bfloat16, "bf16"), (torch.float32, "fp32")):
            model = hf_model(m, vocab, H, heads, kv_heads, F, L, T + new + 4, theta, dt)
            with torch.no_grad():
                o = model.generate(ids, attention_mask=torch.ones_like(ids), do_sample=False, num_beams=1, max_new_tokens=new,
                                   min_new_tokens=new, output_scores=True, return_dict_in_generate=True)
            res[f"ids_{tag}"] = o.sequences.numpy().astype(np.int64)
            sc = torch.stack(o.scores, 1).float()
            top2 = sc.topk(2, -1).values
            res[f"gap_{tag}"] = (top2[..., 0] - top2[..., 1]).numpy()
            if tag == "bf16":
                res["logits0_bf16"] = torch_to_bits(o.scores[0])
        if (res["ids_bf16"] == res["ids_fp32"]).all() and res["gap_bf16"].min() >= min_gap:
            np.savez_compressed(os.path.join(HERE, name + ".npz"), **res)
            print("wrote", name, "seed", seed, "min gap", res["gap_bf16"].min())
            return
    raise RuntimeError("no stable seed for " + name)


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(4)
    run_layer_case("llama_layer_h256", 256, 4, 2, 512, 2, 9, 2, 41, 0.08)
    run_layer_case("llama_layer_h512_d128", 512, 4, 1, 1024, 2, 40, 2, 42, 0.06, theta=500000.0)
    run_layer_case("llama_layer_h256_mha", 256, 8, 8, 768, 3, 17, 2, 43, 0.08)
    run_generate_case("llama_generate_h256", 1024, 256, 4, 2, 512, 3, 2, 12, 6, 700, 0.08, theta=500000.0)
