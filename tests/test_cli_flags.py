"""The CLI keeps the reference's flag surface (run.py:195-215 -> run_generation.py:111-117): same names, types
and defaults; and the summary block prints the reference's four lines (run_generation.py:337-354)."""
import pytest


def test_flag_surface_and_defaults():
    from lia_amd.run_generation import build_parser
    a = build_parser().parse_args([])
    assert (a.prefill_policy, a.decoding_policy, a.no_overlap, a.pin_weight, a.gpu_percentage, a.num_minibatch, a.enable_cxl) == \
        (1, 1, False, False, 0, 1, False)                      # IPEX baseline defaults (lia/modeling_opt.py:1172)
    assert (a.max_new_tokens, a.batch_size, a.num_iter, a.num_warmup, a.input_tokens) == (32, 1, 100, 10, "32")
    b = build_parser().parse_args("--benchmark -m facebook/opt-30b --dtype bfloat16 --ipex --input-tokens 256 --max-new-tokens 32 "
                                  "--batch-size 64 --token-latency --num-iter 10 --num-warmup 2 --greedy --prefill-policy 0 "
                                  "--decoding-policy 2 --gpu-percentage 10 --num-minibatch 2 --pin-weight --enable-cxl --no-overlap".split())
    assert (b.prefill_policy, b.decoding_policy, b.gpu_percentage, b.num_minibatch) == (0, 2, 10, 2)
    assert b.pin_weight and b.enable_cxl and b.no_overlap and b.token_latency and b.greedy and b.ipex and b.benchmark
    assert not a.profile and build_parser().parse_args(["--profile"]).profile        # run_generation.py:103
    with pytest.raises(SystemExit):
        build_parser().parse_args(["--prefill-policy", "x"])


def test_summary_matches_reference_protocol(capsys):
    from lia_amd.run_generation import summarize, synthetic_prompt
    lists = [[1.0, 0.10, 0.12, 0.14], [1.2, 0.11, 0.13, 0.30]]
    res = summarize(total_time=5.0, num_iter=4, num_warmup=2, total_list=lists, batch_size=64)
    out = capsys.readouterr().out
    for line in ("Inference latency: 2.500 sec.", "First token average latency: 1.100 sec.", "Average 2... latency: 0.150 sec.",
                 "P90 2... latency: 0.300 sec.", "P99 2... latency: 0.300 sec."):
        assert line in out
    assert abs(res["prefill_ms"] - 1100.0) < 1e-6 and abs(res["decode_tokens_per_s"] - 64 / 0.15) < 1e-6
    ids = synthetic_prompt(50272, 256, 64)
    assert ids.shape == (64, 256) and (ids == ids[0]).all() and ids[0, 0] == 2 and ids.min() >= 2
