#!/usr/bin/env python3
"""The reference's benchmark matrix through this build's run.py -- one `python run.py <flags>` process per line, the
reference's LIA flags unchanged (README.md:78,101-112; llm/scripts/lia_offline.sh:13-29, lia_online.sh:13-37,
cxl_offloading.sh:13-39).  The reference's deliverable is the set of log files those scripts leave; here every line leaves
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


def run_line(name, flags, prefix, timeout, extra=(), suffix=""):
    js = os.path.join(ROOT, OUTDIR[0], f"{prefix}_{name}{suffix}.json")
    log = os.path.join(ROOT, OUTDIR[0], f"{prefix}_{name}{suffix}.log")
    os.makedirs(os.path.dirname(js), exist_ok=True)
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
