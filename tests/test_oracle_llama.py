"""CPU oracle of the Llama-family layer (BASELINE.json config 4, build-defined) against HF transformers' eager
bf16 Llama executed on CPU (tests/golden/make_golden_llama.py)."""
import glob
import os

import numpy as np
import pytest

import synth
from test_oracle_golden import close

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
LAYER_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLD, "llama_layer_*.npz")))


@pytest.mark.parametrize("name", LAYER_CASES)
def test_llama_layer_matches_hf(oracle, name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    H, heads, kvh, F, B, T, new, seed = [int(v) for v in z["cfg"]]
    m = synth.make_llama_model(seed, 64, H, heads, kvh, F, 1, float(z["w_std"][0]))
    W = m["layers"][0]
    d = H // heads
    cosb, sinb = oracle.rope_tables(T + new + 4, d, float(z["theta"][0]))
    kc = np.zeros((T + new, B, kvh, d), np.uint16)
    vc = np.zeros_like(kc)
    x = synth.make_hidden(seed + 1, B, T, H)
    # HF's output_hidden_states[-1] is taken AFTER the model's final RMSNorm: apply it to the layer output too
    fin = lambda t: oracle.rmsnorm(t, m["final_norm_w"])  # noqa: E731
    y = oracle.llama_layer_forward(W, x, kc, vc, cosb, sinb, 0, heads, kvh)
    close(fin(y), z["prefill_hidden"], atol=0.05, rtol=0.016, frac_exact=0.85)
    for s in range(new):
        xs = synth.make_hidden(seed + 100 + s, B, 1, H)
        ys = oracle.llama_layer_forward(W, xs, kc, vc, cosb, sinb, T + s, heads, kvh)
        close(fin(ys), z[f"dec{s}_hidden"], atol=0.05, rtol=0.016, frac_exact=0.8)
    close(kc, z["kcache"], atol=0.02, rtol=0.008, frac_exact=0.97)     # post-RoPE keys, seq-major
    close(vc, z["vcache"], atol=0.02, rtol=0.008, frac_exact=0.97)


def test_llama_generate_ids_match_hf(oracle):
    z = np.load(os.path.join(GOLD, "llama_generate_h256.npz"))
    vocab, H, heads, kvh, F, L, B, T, new, seed = [int(v) for v in z["cfg"]]
    m = synth.make_llama_model(seed, vocab, H, heads, kvh, F, L, float(z["w_std"][0]))
    ids = synth.make_prompt_ids(seed + 1, B, T, vocab)
    out, lat, logits = oracle.llama_generate(m, ids, new, heads, kvh, float(z["theta"][0]), return_logits=True)
    assert (out == z["ids_bf16"]).all(), (out[0, T:], z["ids_bf16"][0, T:])
    close(logits[0], z["logits0_bf16"], atol=0.06, rtol=0.02)
