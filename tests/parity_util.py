"""Shared checks of the -m gpu parity tests."""
import numpy as np

import synth


def ids_equal_or_near_tie(got_ids, ref_ids, ref_logits, T, what=""):
    """Greedy ids must equal the oracle's; where a row parts from it, the ORACLE's own top-2 logit gap at that step must be
    within two bf16 quanta (a legitimate near-tie), and that ROW is not compared behind its divergent step (its sequence
    differs from there on; the other rows still are).  Returns (first divergent step or None, per-step smallest top-2 gaps) for the test to print."""
    got_ids, ref_ids = np.asarray(got_ids), np.asarray(ref_ids)
    n = got_ids.shape[1] - T
    gaps, quanta = [], []
    for lg in ref_logits:
        f = synth.bf16_bits_to_f32(lg)
        top2 = np.sort(f, -1)[:, -2:]
        gaps.append(top2[:, 1] - top2[:, 0])
        quanta.append(2.0 ** (np.floor(np.log2(np.maximum(np.abs(f).max(-1), 1e-30))) - 7))
    first = None
    alive = np.ones(got_ids.shape[0], bool)          # rows still on the oracle's sequence: a row that parted (near-tie) is not compared further
    for s in range(n):
        bad = np.nonzero(alive & (got_ids[:, T + s] != ref_ids[:, T + s]))[0]
        if bad.size:
            first = s if first is None else first
            for r in bad:
                assert gaps[s][r] <= 2 * quanta[s][r], (f"{what}: row {r} parts from the oracle at step {s} where the oracle's top-2 gap is "
                                                        f"{gaps[s][r]:.4f} (bf16 quantum {quanta[s][r]:.4f}): not a near-tie")
            alive[bad] = False
    return first, [float(g.min()) for g in gaps]


_FULLSIZE = {}


def fullsize_opt30b_case():
    """(fixture, W, x, decode x) of tests/golden/fullsize_layer_opt30b.npz -- outputs of the reference's OPTDecoderLayer_forward at the
    OPT-30B layer shape (make_golden.py).  The 616 M weights are re-derived from the seed (~1 min of numpy), once per process."""
    import os
    if not _FULLSIZE:
        z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fullsize_layer_opt30b.npz"))
        H, heads, F, B, T, new, seed, ident = [int(v) for v in z["cfg"]]
        _FULLSIZE.update(z=z, cfg=(H, heads, F, B, T, new), W=synth.make_layer(seed, H, F, float(z["w_std"][0])),
                         x=synth.make_hidden(seed + 1, B, T, H, bool(ident)), xs=synth.make_hidden(seed + 100, B, 1, H, bool(ident)))
    return _FULLSIZE


def quantum_bound(got, ref, what, min_exact, max_quanta=3.0, within_one=0.999):
    """two correct bf16 implementations of a wide layer: every element within `max_quanta` bf16 quanta of the tensor's largest
    value (+ 2 ulp of its own), `within_one` of them within one quantum, at least `min_exact` bit-identical; prints the rates"""
    a, b = synth.bf16_bits_to_f32(got), synth.bf16_bits_to_f32(ref)
    q = 2.0 ** (np.floor(np.log2(max(float(np.abs(b).max()), 2.0 ** -120))) - 7)
    err = np.abs(a - b)
    frac = float((got == ref).mean())
    print(f"\n{what}: {100 * frac:.2f} % bit-identical, max |err| {err.max():.4g} = {err.max() / q:.2f} quanta of max |ref| {np.abs(b).max():.3g}, "
          f"{100 * (err <= q).mean():.3f} % within one quantum")
    assert (err <= max_quanta * q + 2.0 ** -6 * np.abs(b)).all(), f"{what}: max err {err.max():.4g} > {max_quanta} quanta ({q:.3g})"
    assert (err <= q).mean() >= within_one and frac >= min_exact, f"{what}: {100 * (err <= q).mean():.2f} % within one quantum, {100 * frac:.2f} % identical"
