"""Shared checks of the -m gpu parity tests."""
import numpy as np

import synth


def ids_equal_or_near_tie(got_ids, ref_ids, ref_logits, T, what=""):
    """Greedy ids must equal the oracle's; where a row parts from it, the ORACLE's own top-2 logit gap at that step must be
    within two bf16 quanta (a legitimate near-tie), and nothing is compared behind the first divergent step (the sequences
    differ from there on).  Returns (first divergent step or None, per-step smallest top-2 gaps) for the test to print."""
    got_ids, ref_ids = np.asarray(got_ids), np.asarray(ref_ids)
    n = got_ids.shape[1] - T
    gaps, quanta = [], []
    for lg in ref_logits:
        f = synth.bf16_bits_to_f32(lg)
        top2 = np.sort(f, -1)[:, -2:]
        gaps.append(top2[:, 1] - top2[:, 0])
        quanta.append(2.0 ** (np.floor(np.log2(np.maximum(np.abs(f).max(-1), 1e-30))) - 7))
    first = None
    for s in range(n):
        bad = np.nonzero(got_ids[:, T + s] != ref_ids[:, T + s])[0]
        if bad.size:
            first = s
            for r in bad:
                assert gaps[s][r] <= 2 * quanta[s][r], (f"{what}: row {r} parts from the oracle at step {s} where the oracle's top-2 gap is "
                                                        f"{gaps[s][r]:.4f} (bf16 quantum {quanta[s][r]:.4f}): not a near-tie")
            break
    return first, [float(g.min()) for g in gaps]
