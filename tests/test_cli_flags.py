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


def test_prompt_goes_through_the_checkpoint_directorys_tokenizer(tmp_path, capsys):
    """--prompt (run_generation.py:87,264-285): tokenized with the tokenizer the -m directory ships, one identical row per batch
    entry, "---- Prompt size" printed; without tokenizer files --prompt is an error, and without --prompt the ids stay synthetic"""
    import torch
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast
    from lia_amd import run_generation as rg
    words = ["<pad>", "</s>", "<unk>", "the", "weights", "stay", "on", "host", "and", "stream", "layer", "by"]
    tok = Tokenizer(models.WordLevel({w: i for i, w in enumerate(words)}, unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.Whitespace()
    d = tmp_path / "ckpt"
    d.mkdir()
    PreTrainedTokenizerFast(tokenizer_object=tok, unk_token="<unk>", pad_token="<pad>", eos_token="</s>").save_pretrained(str(d))
    t = rg.load_tokenizer(str(d))
    assert t is not None and rg.load_tokenizer("facebook/opt-30b") is None and rg.load_tokenizer(str(tmp_path)) is None
    args = rg.build_parser().parse_args(["-m", str(d), "--batch-size", "3", "--prompt", "the weights stay on host and stream layer by layer"])
    ids = rg.prompt_input_ids(args, 64, t)
    assert ids.dtype == torch.int64 and ids.shape == (3, 10) and ids[0].tolist() == [3, 4, 5, 6, 7, 8, 9, 10, 11, 10]
    assert (ids == ids[0]).all() and "---- Prompt size: 10" in capsys.readouterr().out
    assert t.batch_decode(ids, skip_special_tokens=True)[0] == "the weights stay on host and stream layer by layer"
    with pytest.raises(SystemExit, match="tokenizer"):
        rg.prompt_input_ids(args, 64, None)
    with pytest.raises(SystemExit, match="vocabulary"):
        rg.prompt_input_ids(args, 8, t)                            # the model's embedding table is smaller than the tokenizer's ids
    plain = rg.build_parser().parse_args(["-m", str(d), "--batch-size", "2", "--input-tokens", "16"])
    assert plain.prompt is None and torch.equal(rg.prompt_input_ids(plain, 64, t), rg.synthetic_prompt(64, 16, 2))


def test_matrix_lines_parse_through_the_harness_flag_surface():
    """tools/run_matrix.py: 53 lines (README example + lia_offline.sh 8 + lia_online.sh 12 + cxl_offloading.sh 12 + the CPU-only baseline
    of ipex_offline.sh 8 / ipex_online.sh 12), every one a flag list the harness's own parser accepts; the two ragged ones (1150 / 3,
    1 / 2) are there"""
    import importlib.util
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("run_matrix", os.path.join(root, "tools", "run_matrix.py"))
    rm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rm)
    sys.path.insert(0, os.path.join(root, "isca-2025-lia_amd"))
    from lia_amd.run_generation import build_parser
    lines = rm.lines()
    assert len(lines) == 53 and len({n for n, _ in lines}) == 53
    by = {n: build_parser().parse_args(f) for n, f in lines}
    assert sum(n.startswith("offline_") for n in by) == 8 and sum(n.startswith("online_") for n in by) == 12 and sum(n.startswith("cxl_") for n in by) == 12
    a = by["cxl_opt30b_32_128_b1150_p02_g0_cxl"]
    assert (a.batch_size, a.num_minibatch, a.enable_cxl, a.pin_weight, a.prefill_policy, a.decoding_policy) == (1150, 3, True, True, 0, 2)
    b = by["offline_opt175b_32_32_b1_p01_g9"]
    assert (b.batch_size, b.num_minibatch, b.pin_weight, b.gpu_percentage, b.init, b.model_id) == (1, 2, False, 9, "uniform01", "opt-175b")
    r = by["readme_opt30b_256_32_b64_p01_g10_cxl"]
    assert (r.num_iter, r.num_warmup, r.input_tokens, r.max_new_tokens, r.gpu_percentage, r.num_minibatch) == (10, 2, "256", 32, 10, 2)
    assert sum(n.startswith("ipexoffline_") for n in by) == 8 and sum(n.startswith("ipexonline_") for n in by) == 12
    assert all((v.prefill_policy, v.decoding_policy, v.gpu_percentage, v.pin_weight) == (1, 1, 0, False) for n, v in by.items() if n.startswith("ipex"))
    assert all(v.benchmark and v.token_latency and v.greedy and v.ipex and v.dtype == "bfloat16" for v in by.values())
