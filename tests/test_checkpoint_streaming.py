"""CPU tests of the streaming checkpoint conversion (lia_amd.checkpoint): a HF directory is walked one
decoder layer at a time -- safetensors through safe_open per tensor, .bin shards through pytorch_model.bin.index.json (the layout
of the reference's own OPT-175B dummy directory, llm/utils/opt-weight-gen.py:61-69) -- so the peak host memory of a conversion
is about one layer, not the checkpoint (r05 held every shard in one dict plus a copy of every tensor)."""
import json
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "isca-2025-lia_amd")

WRITER = textwrap.dedent("""
    import json, os, sys
    import numpy as np, torch
    path, kind = sys.argv[1], sys.argv[2]
    H, F, L, heads, vocab, max_pos = 1024, 4096, 12, 8, 256, 64
    os.makedirs(path, exist_ok=True)
    json.dump(dict(model_type="opt", hidden_size=H, ffn_dim=F, num_hidden_layers=L, num_attention_heads=heads, vocab_size=vocab,
                   max_position_embeddings=max_pos, do_layer_norm_before=True, word_embed_proj_dim=H, torch_dtype="bfloat16"),
              open(os.path.join(path, "config.json"), "w"))
    g = torch.Generator().manual_seed(5)
    rnd = lambda *s: (0.02 * torch.randn(*s, generator=g)).to(torch.bfloat16)
    hf = {"self_attn_layer_norm": (H,), "self_attn.q_proj": (H, H), "self_attn.k_proj": (H, H), "self_attn.v_proj": (H, H),
          "self_attn.out_proj": (H, H), "final_layer_norm": (H,), "fc1": (F, H), "fc2": (H, F)}
    shards, weight_map = [], {}
    head = {"model.decoder.embed_tokens.weight": rnd(vocab, H), "model.decoder.embed_positions.weight": rnd(max_pos + 2, H),
            "model.decoder.final_layer_norm.weight": rnd(H), "model.decoder.final_layer_norm.bias": rnd(H)}
    per = 3                                   # layers per shard
    for s0 in range(0, L, per):
        sd = dict(head) if s0 == 0 else {}
        for i in range(s0, s0 + per):
            for name, shp in hf.items():
                sd[f"model.decoder.layers.{i}.{name}.weight"] = rnd(*shp)
                sd[f"model.decoder.layers.{i}.{name}.bias"] = rnd(shp[0])
        n = len(shards) + 1
        if kind == "safetensors":
            from safetensors.torch import save_file
            fn = f"model-{n:05d}-of-{L // per:05d}.safetensors"
            save_file(sd, os.path.join(path, fn))
        else:
            fn = f"pytorch_model-{n:05d}-of-{L // per:05d}.bin"
            torch.save(sd, os.path.join(path, fn))
        shards.append(fn)
        for k in sd:
            weight_map[k] = fn
    if kind == "bin":
        json.dump({"metadata": {}, "weight_map": weight_map}, open(os.path.join(path, "pytorch_model.bin.index.json"), "w"))
    # a checksum of layer 7's fc1 weight, for the reader to compare
    w = sd_last = None
    print(json.dumps({"layer_bytes": 2 * (4 * H * H + 2 * H * F + 9 * H + F)}))
""")

READER = textwrap.dedent("""
    import json, os, sys
    sys.path.insert(0, sys.argv[2])
    import numpy as np, torch
    from lia_amd import checkpoint

    def stat(key):
        for line in open("/proc/self/status"):
            if line.startswith(key):
                return int(line.split()[1]) * 1024

    def hwm():
        return stat("VmHWM")
    src, pre, layers = checkpoint.iter_hf_opt_layers(sys.argv[1])
    head = checkpoint.opt_head_numpy(src, pre)
    base, anon0, anon_peak = hwm(), stat("RssAnon"), 0
    n, acc, shapes = 0, 0, None
    for i, lw in layers:
        n += 1
        anon_peak = max(anon_peak, stat("RssAnon") - anon0)          # with one layer dict alive: the conversion's own allocations
        acc ^= int(lw["fc1_w"].view(np.uint16)[::997].astype(np.uint64).sum())   # touch the data
        shapes = {k: list(v.shape) for k, v in lw.items()}
        del lw
    print(json.dumps({"layers": n, "peak_growth": hwm() - base, "anon_growth": anon_peak, "shard_loads": src.shard_loads, "kind": src.kind, "shapes": shapes, "acc": acc}))
""")


@pytest.mark.parametrize("kind", ["safetensors", "bin"])
def test_conversion_walks_one_layer_at_a_time(tmp_path, kind):
    ck = str(tmp_path / "opt-12l")
    w = subprocess.run([sys.executable, "-c", WRITER, ck, kind], capture_output=True, text=True, timeout=600)
    assert w.returncode == 0, w.stderr[-2000:]
    layer_bytes = json.loads(w.stdout.strip().splitlines()[-1])["layer_bytes"]            # 25.2 MB; the directory holds 12 of them
    r = subprocess.run([sys.executable, "-c", READER, ck, PKG], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["layers"] == 12 and out["kind"] == kind
    assert out["shapes"]["fc1_w"] == [4096, 1024] and out["shapes"]["fc2_w"] == [1024, 4096] and out["shapes"]["q_b"] == [1024]
    # the conversion's own (anonymous) memory: the layer being packed (+ the tensor being read) -- under two layers, against twelve
    # for a load-everything conversion.  The resident-set high-water mark also counts the file pages of the shard that is mapped at
    # the moment (safe_open and torch.load(mmap=True) both map; those pages are page cache, reclaimable): one three-layer shard more
    assert out["anon_growth"] < 2 * layer_bytes, out
    assert out["peak_growth"] < (3 + 2) * layer_bytes, out
    if kind == "bin":
        assert out["shard_loads"] == 4, out                   # every shard opened exactly once by the layer-order walk


def test_tensor_source_without_an_index_and_missing_files(tmp_path):
    import torch
    sys.path.insert(0, PKG)
    from lia_amd.checkpoint import TensorSource
    with pytest.raises(FileNotFoundError):
        TensorSource(str(tmp_path))
    torch.save({"a.weight": torch.ones(4, 4, dtype=torch.bfloat16), "b.bias": torch.zeros(4, dtype=torch.bfloat16)},
               str(tmp_path / "pytorch_model.bin"))
    src = TensorSource(str(tmp_path))
    assert src.kind == "bin" and set(src.names()) == {"a.weight", "b.bias"} and "a.weight" in src and "c" not in src
    assert src.get("a.weight").shape == (4, 4) and src.shard_loads == 1
    src.close()


def test_llama_directory_detection(tmp_path):
    sys.path.insert(0, PKG)
    from lia_amd import checkpoint, run_generation
    d = tmp_path / "tiny-llama"
    d.mkdir()
    json.dump(dict(architectures=["LlamaForCausalLM"], model_type="llama", hidden_size=256, num_attention_heads=4, num_key_value_heads=2,
                   intermediate_size=512, num_hidden_layers=2, vocab_size=128, max_position_embeddings=64, rope_theta=10000.0,
                   rms_norm_eps=1e-5), open(d / "config.json", "w"))
    assert checkpoint.is_llama_dir(str(d))
    sh = checkpoint.llama_shape_of(str(d))
    assert (sh.hidden, sh.heads, sh.kv_heads, sh.ffn, sh.layers, sh.vocab, sh.max_pos, sh.rope_theta) == (256, 4, 2, 512, 2, 128, 64, 10000.0)
    a = run_generation.build_parser().parse_args(["-m", str(d)])
    assert run_generation.is_llama(a) and run_generation.model_shape(a) == sh
    for name, want in (("meta-llama/Llama-3-8B", True), ("meta-llama/Meta-Llama-3-8B", True), ("facebook/opt-30b", False), ("opt-175b", False)):
        assert run_generation.is_llama(run_generation.build_parser().parse_args(["-m", name])) is want
    json.dump(dict(architectures=["LlamaForCausalLM"], model_type="llama", hidden_size=256, num_attention_heads=4, intermediate_size=512,
                   num_hidden_layers=2, vocab_size=128, rope_scaling={"rope_type": "llama3", "factor": 8.0}), open(d / "config.json", "w"))
    with pytest.raises(ValueError, match="rope_scaling"):
        checkpoint.llama_shape_of(str(d))
    o = tmp_path / "opt"
    o.mkdir()
    json.dump(dict(model_type="opt", architectures=["OPTForCausalLM"]), open(o / "config.json", "w"))
    assert not checkpoint.is_llama_dir(str(o)) and not checkpoint.is_llama_dir(str(tmp_path / "nothing"))


def test_llama_layer_of_a_hf_directory_lands_in_the_packed_layout(tmp_path):
    """checkpoint.llama_layer_numpy + LiaLlamaModel._pack_numpy without a GPU: the nine tensors of a HF Llama layer (safetensors, two
    shards) end up at lia_llama_pack_offsets' positions, gate.w / up.w interleaved in blocks of 32 rows (lia_llama_desc.gu_block)"""
    import torch
    from safetensors.torch import save_file
    sys.path.insert(0, PKG)
    from lia_amd import checkpoint
    from lia_amd.llama import GU_BLOCK, LLAMA_TENSORS, LiaLlamaModel
    H, heads, kvh, F, L, vocab = 256, 4, 2, 512, 2, 64
    d = tmp_path / "tiny-llama"
    d.mkdir()
    json.dump(dict(architectures=["LlamaForCausalLM"], model_type="llama", hidden_size=H, num_attention_heads=heads, num_key_value_heads=kvh,
                   intermediate_size=F, num_hidden_layers=L, vocab_size=vocab, max_position_embeddings=64, rope_theta=10000.0), open(d / "config.json", "w"))
    g = torch.Generator().manual_seed(3)
    KD = kvh * (H // heads)
    dims = {"input_layernorm": (H,), "self_attn.q_proj": (H, H), "self_attn.k_proj": (KD, H), "self_attn.v_proj": (KD, H), "self_attn.o_proj": (H, H),
            "post_attention_layernorm": (H,), "mlp.gate_proj": (F, H), "mlp.up_proj": (F, H), "mlp.down_proj": (H, F)}
    sd = [{}, {}]
    for i in range(L):
        for name, shp in dims.items():
            sd[i][f"model.layers.{i}.{name}.weight"] = torch.randn(*shp, generator=g).to(torch.bfloat16)
    sd[0]["model.embed_tokens.weight"] = torch.randn(vocab, H, generator=g).to(torch.bfloat16)
    sd[1]["model.norm.weight"] = torch.ones(H, dtype=torch.bfloat16)
    for n, part in enumerate(sd):
        save_file(part, str(d / f"model-{n + 1:05d}-of-00002.safetensors"))
    shape = checkpoint.llama_shape_of(str(d))
    src = checkpoint.TensorSource(str(d))
    assert "lm_head.weight" not in src                                   # a tied checkpoint: load_hf_llama falls back to the embedding
    model = LiaLlamaModel(shape)
    bits = lambda t: t.contiguous().view(torch.int16).numpy().view(np.uint16)      # noqa: E731
    for i, st in enumerate(model.layers):
        lw = checkpoint.llama_layer_numpy(src, i)
        assert set(lw) == set(LLAMA_TENSORS)
        model._pack_numpy(st, lw)
        flat = st._np.view(np.uint16)
        off = {n: model.offsets[k] // 2 for k, n in enumerate(LLAMA_TENSORS)}
        hf = {"in_norm_w": "input_layernorm", "q_w": "self_attn.q_proj", "k_w": "self_attn.k_proj", "v_w": "self_attn.v_proj", "o_w": "self_attn.o_proj",
              "post_norm_w": "post_attention_layernorm", "down_w": "mlp.down_proj"}
        for short, name in hf.items():
            want = bits(sd[i][f"model.layers.{i}.{name}.weight"]).reshape(-1)
            assert (flat[off[short]: off[short] + want.size] == want).all(), (i, short)
        gate, up = bits(sd[i][f"model.layers.{i}.mlp.gate_proj.weight"]), bits(sd[i][f"model.layers.{i}.mlp.up_proj.weight"])
        gu = flat[off["gate_w"]: off["gate_w"] + 2 * F * H].reshape(F // GU_BLOCK, 2, GU_BLOCK, H)
        assert (gu[:, 0].reshape(F, H) == gate).all() and (gu[:, 1].reshape(F, H) == up).all()
    src.close()
