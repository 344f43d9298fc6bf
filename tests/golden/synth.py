"""Deterministic synthetic tensors shared by the golden-vector generator and the tests.

Everything is drawn from numpy's frozen legacy MT19937 stream (np.random.RandomState), whose
output is guaranteed stable across numpy versions, so a fixture only has to record a seed and the
expected outputs -- never the weights themselves.

bf16 values travel as uint16 bit patterns (numpy has no bf16); rounding is round-to-nearest-even,
the same rounding torch's .to(torch.bfloat16) applies.
"""
import numpy as np

# Fixed order of the 16 per-layer tensors -- the order of create_buffer()
# (reference lia/modeling_opt.py:90-126) that decoder.py / attentions.py index into.
LAYER_TENSORS = (
    "ln1_w", "ln1_b", "q_w", "q_b", "k_w", "k_b", "v_w", "v_b",
    "out_w", "out_b", "ln2_w", "ln2_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b",
)


def f32_to_bf16_bits(x):
    """float32 ndarray -> uint16 bf16 bit patterns, round-to-nearest-even (NaN kept quiet)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32)
    rounded = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)
    nan = np.isnan(x)
    if nan.any():
        rounded = np.where(nan, np.uint16(0x7FC0), rounded)
    return rounded


def bf16_bits_to_f32(b):
    b = np.ascontiguousarray(b, dtype=np.uint16)
    return (b.astype(np.uint32) << 16).view(np.float32)


def layer_shapes(H, F):
    return {
        "ln1_w": (H,), "ln1_b": (H,), "q_w": (H, H), "q_b": (H,), "k_w": (H, H), "k_b": (H,),
        "v_w": (H, H), "v_b": (H,), "out_w": (H, H), "out_b": (H,), "ln2_w": (H,), "ln2_b": (H,),
        "fc1_w": (F, H), "fc1_b": (F,), "fc2_w": (H, F), "fc2_b": (H,),
    }


def make_layer(seed, H, F, w_std=0.02):
    """One decoder layer's 16 tensors as bf16 bit patterns, row-major [N, K] linears.

    Linear/bias ~ normal(0, w_std) (HF OPT _init_weights uses 0.02 and zero bias; biases and LN
    parameters are perturbed here so that every term of the layer is exercised)."""
    rs = np.random.RandomState(seed)
    out = {}
    for name, shape in layer_shapes(H, F).items():
        if name in ("ln1_w", "ln2_w"):
            t = 1.0 + 0.1 * rs.standard_normal(shape)
        elif name.endswith("_b"):
            t = 0.05 * rs.standard_normal(shape)
        else:
            n = int(np.prod(shape))
            if n > (1 << 22):
                # big matrices (the full-size fixtures: 616 M values per OPT-30B layer) in chunks -- the SAME stream of normals
                # (RandomState.standard_normal is sequential), without multi-GB float64 temporaries
                bits = np.empty(n, np.uint16)
                for o in range(0, n, 1 << 22):
                    m = min(1 << 22, n - o)
                    bits[o:o + m] = f32_to_bf16_bits((w_std * rs.standard_normal(m)).astype(np.float32))
                out[name] = bits.reshape(shape)
                continue
            t = w_std * rs.standard_normal(shape)
        out[name] = f32_to_bf16_bits(t.astype(np.float32))
    return out


def make_hidden(seed, B, T, H, identical_rows=False):
    rs = np.random.RandomState(seed)
    if identical_rows:
        x = np.tile(rs.standard_normal((1, T, H)), (B, 1, 1))
    else:
        x = rs.standard_normal((B, T, H))
    return f32_to_bf16_bits(x.astype(np.float32))


def make_prompt_ids(seed, B, T, vocab):
    """run_generation.py:285 batches one prompt replicated B times -> identical rows; BOS=2 first."""
    rs = np.random.RandomState(seed)
    row = rs.randint(4, vocab, size=(T,)).astype(np.int64)
    row[0] = 2
    return np.tile(row[None, :], (B, 1))


def make_model(seed, vocab, max_pos, H, F, L, w_std=0.02):
    """Whole tiny OPT: embed_tokens [vocab,H] (tied lm_head), embed_positions [max_pos+2,H],
    L layers, final LN. All bf16 bit patterns."""
    rs = np.random.RandomState(seed)
    m = {
        "embed_tokens": f32_to_bf16_bits((w_std * rs.standard_normal((vocab, H))).astype(np.float32)),
        "embed_positions": f32_to_bf16_bits((w_std * rs.standard_normal((max_pos + 2, H))).astype(np.float32)),
        "final_ln_w": f32_to_bf16_bits((1.0 + 0.1 * rs.standard_normal((H,))).astype(np.float32)),
        "final_ln_b": f32_to_bf16_bits((0.05 * rs.standard_normal((H,))).astype(np.float32)),
        "layers": [make_layer(seed * 1000 + 17 * (i + 1), H, F, w_std) for i in range(L)],
    }
    return m


# ---- Llama family (config 4) ------------------------------------------------------------------------------------
LLAMA_TENSORS = ("in_norm_w", "q_w", "k_w", "v_w", "o_w", "post_norm_w", "gate_w", "up_w", "down_w")


def make_llama_layer(seed, H, heads, kv_heads, F, w_std=0.02):
    d = H // heads
    rs = np.random.RandomState(seed)
    shapes = {"in_norm_w": (H,), "q_w": (H, H), "k_w": (kv_heads * d, H), "v_w": (kv_heads * d, H), "o_w": (H, H),
              "post_norm_w": (H,), "gate_w": (F, H), "up_w": (F, H), "down_w": (H, F)}
    out = {}
    for n in LLAMA_TENSORS:
        if n.endswith("norm_w"):
            t = 1.0 + 0.1 * rs.standard_normal(shapes[n])
        else:
            t = w_std * rs.standard_normal(shapes[n])
        out[n] = f32_to_bf16_bits(t.astype(np.float32))
    return out


def make_llama_model(seed, vocab, H, heads, kv_heads, F, L, w_std=0.02):
    rs = np.random.RandomState(seed)
    return {
        "embed_tokens": f32_to_bf16_bits((w_std * rs.standard_normal((vocab, H))).astype(np.float32)),
        "lm_head": f32_to_bf16_bits((w_std * rs.standard_normal((vocab, H))).astype(np.float32)),
        "final_norm_w": f32_to_bf16_bits((1.0 + 0.1 * rs.standard_normal((H,))).astype(np.float32)),
        "layers": [make_llama_layer(seed * 1000 + 31 * (i + 1), H, heads, kv_heads, F, w_std) for i in range(L)],
    }
