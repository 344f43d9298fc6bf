#!/usr/bin/env python3
"""The reference's benchmark matrix through this build's run.py -- one `python run.py <flags>` process per line, the
reference's LIA flags unchanged (README.md:78,101-112; llm/scripts/lia_offline.sh:13-29, lia_online.sh:13-37,
cxl_offloading.sh:13-39) and, as run_performance.sh does, the CPU-only baseline of the same shapes (ipex_offline.sh / ipex_online.sh:
policies 1 / 1, gpu% 0 -- here this build's host path on the box's cores, `ipexoffline_*` / `ipexonline_*`).  The reference's deliverable is the set of log files those scripts leave; here every line leaves
results/<prefix>_<name>.json (lia_amd.run_generation --result-json: prefill ms, decode tokens/s, the weight stream's share of the
link, which resource dominated the profiled warm-up iteration, the host-memory peak, the planner's pick for the same line) and
results/<prefix>_<name>.log (the harness output the reference's scripts redirect to their .log files).  A line this box cannot
hold is recorded with the reason (status "refused: memory") -- nothing is skipped silently.

    python tools/run_matrix.py --list
    python tools/run_matrix.py --only 'offline_opt30b|readme' [--auto-plan-too] [--timeout 1500]

Differences from the scripts, all forced by the box: no `OMP_NUM_THREADS=40 numactl -m 0 -C 0-39` prefix (the harness pins itself
to the GPU's NUMA node and sizes its OpenMP team from the cgroup quota, lia_amd.hostinfo); `-m /home/storage/opt-175b/` (a
directory of torch.rand dummy weights, README.md section 5.1) is `-m opt-175b --init uniform01` -- the same recipe drawn in the
process; prompts are synthetic ids (no prompt.json offline)."""
import argparse
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUTDIR = ["results"]

# (script, model, input tokens, new tokens, batch, prefill policy, decode policy, minibatches, gpu %, --pin-weight, --enable-cxl, iters, warm-up)
M30, M175 = "facebook/opt-30b", "opt-175b"
TABLE = [
    ("readme",  M30, 256, 32, 64, 0, 1, 2, 10, True, True, 10, 2),          # README.md:78, the quick example
    # llm/scripts/lia_offline.sh:13-29
    ("offline", M30, 32, 32, 64, 0, 1, 1, 50, True, False, 2, 1),
    ("offline", M30, 2016, 32, 64, 0, 1, 8, 0, True, False, 2, 1),
    ("offline", M30, 32, 256, 64, 0, 1, 1, 40, True, False, 2, 1),
    ("offline", M30, 1792, 256, 64, 0, 1, 8, 0, True, False, 2, 1),
    ("offline", M30, 32, 32, 900, 0, 2, 2, 0, True, False, 2, 1),
    ("offline", M30, 32, 256, 900, 0, 2, 2, 0, True, False, 2, 1),
    ("offline", M175, 32, 32, 1, 0, 1, 2, 9, False, False, 2, 1),
    ("offline", M175, 32, 256, 1, 0, 1, 2, 7, False, False, 2, 1),
    # llm/scripts/lia_online.sh:13-37
    ("online", M30, 32, 32, 1, 1, 1, 1, 66, True, False, 2, 1),
    ("online", M30, 256, 32, 1, 1, 1, 1, 64, True, False, 2, 1),
    ("online", M30, 2016, 32, 1, 0, 1, 1, 58, True, False, 2, 1),
    ("online", M30, 32, 256, 1, 1, 1, 1, 64, True, False, 2, 1),
    ("online", M30, 256, 256, 1, 1, 1, 1, 62, True, False, 2, 1),
    ("online", M30, 1792, 256, 1, 0, 1, 1, 58, True, False, 2, 1),
    ("online", M175, 32, 32, 1, 1, 1, 1, 12, False, False, 2, 1),
    ("online", M175, 256, 32, 1, 1, 1, 1, 12, False, False, 2, 1),
    ("online", M175, 2016, 32, 1, 0, 1, 1, 8, False, False, 2, 1),
    ("online", M175, 32, 256, 1, 1, 1, 1, 12, False, False, 2, 1),
    ("online", M175, 256, 256, 1, 1, 1, 1, 10, False, False, 2, 1),
    ("online", M175, 1792, 256, 1, 0, 1, 1, 9, False, False, 2, 1),
    # llm/scripts/cxl_offloading.sh:13-39
    ("cxl", M30, 32, 32, 900, 0, 2, 2, 0, True, False, 2, 1),
    ("cxl", M30, 32, 64, 900, 0, 2, 2, 0, True, False, 2, 1),
    ("cxl", M30, 32, 128, 900, 0, 2, 2, 0, True, False, 2, 1),
    ("cxl", M30, 32, 256, 900, 0, 2, 2, 0, True, False, 2, 1),
    ("cxl", M30, 32, 32, 900, 0, 2, 2, 0, True, True, 2, 1),
    ("cxl", M30, 32, 64, 900, 0, 2, 2, 0, True, True, 2, 1),
    ("cxl", M30, 32, 128, 900, 0, 2, 2, 0, True, True, 2, 1),
    ("cxl", M30, 32, 256, 900, 0, 2, 2, 0, True, True, 2, 1),
    ("cxl", M30, 32, 32, 1580, 0, 2, 4, 0, True, True, 2, 1),
    ("cxl", M30, 32, 64, 1350, 0, 2, 3, 0, True, True, 2, 1),
    ("cxl", M30, 32, 128, 1150, 0, 2, 3, 0, True, True, 2, 1),
    ("cxl", M30, 32, 256, 1050, 0, 2, 3, 0, True, True, 2, 1),
    # the CPU-only baseline of the same shapes: llm/scripts/ipex_offline.sh:13-29 and ipex_online.sh:13-37 (run_performance.sh runs
    # them beside the LIA lines): policies 1 / 1, gpu% 0, no --pin-weight -- every layer on the host cores
    ("ipexoffline", M30, 32, 32, 64, 1, 1, 1, 0, False, False, 2, 1),
    ("ipexoffline", M30, 2016, 32, 64, 1, 1, 1, 0, False, False, 2, 1),
    ("ipexoffline", M30, 32, 256, 64, 1, 1, 1, 0, False, False, 2, 1),
    ("ipexoffline", M30, 1792, 256, 64, 1, 1, 1, 0, False, False, 2, 1),
    ("ipexoffline", M30, 32, 32, 900, 1, 1, 1, 0, False, False, 2, 1),
    ("ipexoffline", M30, 32, 256, 900, 1, 1, 1, 0, False, False, 2, 1),
    ("ipexoffline", M175, 32, 32, 1, 1, 1, 1, 0, False, False, 2, 1),
    ("ipexoffline", M175, 32, 256, 1, 1, 1, 1, 0, False, False, 2, 1),
    ("ipexonline", M30, 32, 32, 1, 1, 1, 1, 0, False, False, 2, 1),
    ("ipexonline", M30, 256, 32, 1, 1, 1, 1, 0, False, False, 2, 1),
    ("ipexonline", M30, 2016, 32, 1, 1, 1, 1, 0, False, False, 2, 1),
    ("ipexonline", M30, 32, 256, 1, 1, 1, 1, 0, False, False, 2, 1),
    ("ipexonline", M30, 256, 256, 1, 1, 1, 1, 0, False, False, 2, 1),
    ("ipexonline", M30, 1792, 256, 1, 1, 1, 1, 0, False, False, 2, 1),
    ("ipexonline", M175, 32, 32, 1, 1, 1, 1, 0, False, False, 2, 1),
    ("ipexonline", M175, 256, 32, 1, 1, 1, 1, 0, False, False, 2, 1),
    ("ipexonline", M175, 2016, 32, 1, 1, 1, 1, 0, False, False, 2, 1),
    ("ipexonline", M175, 32, 256, 1, 1, 1, 1, 0, False, False, 2, 1),
    ("ipexonline", M175, 256, 256, 1, 1, 1, 1, 0, False, False, 2, 1),
    ("ipexonline", M175, 1792, 256, 1, 1, 1, 1, 0, False, False, 2, 1),
]


def lines():
    out = []
    for (script, model, t_in, t_new, bs, pp, dp, mb, pct, pin, cxl, iters, warm) in TABLE:
        short = "opt175b" if model == M175 else "opt30b"
        name = f"{script}_{short}_{t_in}_{t_new}_b{bs}_p{pp}{dp}_g{pct}" + ("_cxl" if cxl else "")
        flags = ["--benchmark", "-m", model] + (["--init", "uniform01"] if model == M175 else []) + \
                ["--dtype", "bfloat16", "--ipex", "--input-tokens", str(t_in), "--max-new-tokens", str(t_new), "--batch-size", str(bs),
                 "--token-latency", "--num-iter", str(iters), "--num-warmup", str(warm), "--greedy", "--prefill-policy", str(pp),
                 "--decoding-policy", str(dp), "--num-minibatch", str(mb), "--gpu-percentage", str(pct)] + \
                (["--pin-weight"] if pin else []) + (["--enable-cxl"] if cxl else [])
        out.append((name, flags))
    return out


HOST_FLOPS = 8.0e12          # what 16 Zen 5 cores sustain with vdpbf16ps in lia_host_linear at M = 64 (128 GB/s of weights: LABNOTES r06)
HOST_PREFILL_FLOPS = 4.5e12  # ... and in a policy-1 prefill (M = 2048: 27.7 s for 48 layers, results/r06_matrix_ipexoffline_opt30b_32_32_b64_p11_g0.json)
HOST_STREAM = 435e9          # bytes/s of weights a batch-1 host layer reads (7.36 tokens/s over 59 GB, ipexonline_opt30b_32_32_b1)


def host_time_estimate(flags):
    """seconds a line with the PREFILL on the host cores (policy 1) would take here: iterations x (prefill flops + new tokens x the
    larger of the decode step's flops and its weight bytes).  The reference ran these lines on 40 Sapphire Rapids cores with AMX."""
    a = dict(zip(flags[::1], flags[1::1]))
    if a.get("--prefill-policy") != "1":
        return 0.0
    params = {"facebook/opt-30b": 29.6e9, "opt-175b": 174e9}[a["-m"]] * (1.0 - int(a["--gpu-percentage"]) / 100.0)
    B, T, new, iters = int(a["--batch-size"]), int(a["--input-tokens"]), int(a["--max-new-tokens"]), int(a["--num-iter"])
    return iters * (2.0 * B * T * params / HOST_PREFILL_FLOPS + new * max(2.0 * B * params / HOST_FLOPS, 2.0 * params / HOST_STREAM))


def run_line(name, flags, prefix, timeout, extra=(), suffix=""):
    js = os.path.join(ROOT, OUTDIR[0], f"{prefix}_{name}{suffix}.json")
    log = os.path.join(ROOT, OUTDIR[0], f"{prefix}_{name}{suffix}.log")
    os.makedirs(os.path.dirname(js), exist_ok=True)
    est = host_time_estimate(flags) if not extra else 0.0
    if est > 0.8 * timeout:
        # not started: the box's 16 cores would need longer than the line's time limit for the host-side prefill / decode alone
        rec = {"status": "not run: time", "flags_cmdline": flags, "estimated_s": round(est),
               "reason": f"policy-1 prefill and decode on this box's 16 host cores: ~{est:.0f} s estimated (prefill {HOST_PREFILL_FLOPS / 1e12:.1f} TFLOP/s, "
                         f"decode {HOST_FLOPS / 1e12:.0f} TFLOP/s or {HOST_STREAM / 1e9:.0f} GB/s: measured on the lines that ran), limit {timeout} s per line"}
        json.dump(rec, open(js, "w"), indent=1)
        print(f"{name}{suffix}: not run -- {rec['reason']}", flush=True)
        return rec
    cmd = [sys.executable, os.path.join(ROOT, "run.py")] + flags + list(extra) + ["--result-json", js]
    t0 = time.time()
    if os.path.exists(js):
        os.remove(js)
    with open(log, "w") as f:
        f.write("$ python run.py " + " ".join(flags + list(extra)) + "\n")
        f.flush()
        try:
            rc = subprocess.run(cmd, stdout=f, stderr=subprocess.STDOUT, timeout=timeout, cwd=ROOT).returncode
        except subprocess.TimeoutExpired:
            rc = "timeout"
    dt = time.time() - t0
    # keep the logs small: the head (flags, placement) and the tail (iterations, summary)
    txt = open(log, errors="replace").read().splitlines()
    if len(txt) > 80:
        txt = txt[:20] + [f"... ({len(txt) - 60} lines cut) ..."] + txt[-40:]
    open(log, "w").write("\n".join(ln[:600] for ln in txt) + "\n")
    if not os.path.exists(js):       # the process died before it could write its record (a timeout, a crash): say so
        json.dump({"status": f"failed: exit {rc}", "flags_cmdline": flags + list(extra), "wall_s": round(dt, 1),
                   "reason": "\n".join(txt[-6:])}, open(js, "w"), indent=1)
    rec = json.load(open(js))
    r = rec.get("result") or {}
    print(f"{name}{suffix}: {rec.get('status')} in {dt:.0f} s" +
          (f" -- prefill {r.get('prefill_ms', float('nan')):.1f} ms, decode {r.get('decode_tokens_per_s', float('nan')):.2f} tokens/s" if r else
           f" -- {str(rec.get('reason'))[:200]}"), flush=True)
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--list", action="store_true")
    ap.add_argument("--only", default=".", help="regular expression over the line names")
    ap.add_argument("--prefix", default="r06_matrix")
    ap.add_argument("--outdir", default="results", help="directory under the repo root (the GPU box only brings gpurun_out/ back)")
    ap.add_argument("--timeout", type=int, default=1800, help="seconds per line")
    ap.add_argument("--auto-plan-too", action="store_true", help="run every selected line a second time with --auto-plan (the planner's flags instead of the hand-picked ones)")
    ap.add_argument("--auto-plan-only", action="store_true")
    ap.add_argument("--budget-s", type=int, default=0, help="stop starting new lines once this many seconds have passed")
    a = ap.parse_args()
    OUTDIR[0] = a.outdir
    t0 = time.time()
    for name, flags in lines():
        if not re.search(a.only, name):
            continue
        if a.list:
            print(name, "|", "python run.py", " ".join(flags))
            continue
        if a.budget_s and time.time() - t0 > a.budget_s:
            print(f"{name}: not started (budget of {a.budget_s} s used up)")
            continue
        if not a.auto_plan_only:
            run_line(name, flags, a.prefix, a.timeout)
        if a.auto_plan_too or a.auto_plan_only:
            run_line(name, flags, a.prefix, a.timeout, extra=["--auto-plan"], suffix="_autoplan")


if __name__ == "__main__":
    main()
