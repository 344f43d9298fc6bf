#!/usr/bin/env python3
"""results/r06_matrix_*.json (tools/run_matrix.py) -> the markdown table of BASELINE.md section 5: one row per line of the reference's
benchmark matrix -- prefill ms, decode tokens/s, the weight stream's share of the 63 GB/s link, the resource that dominated the
profiled warm-up iteration, host memory of the run, the planner's pick for the same line (and the measured --auto-plan run where
there is one).   python tools/matrix_table.py [--dir results] [--prefix r06_matrix]"""
import argparse
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import run_matrix  # noqa: E402


def fmt(v, f="{:.1f}"):
    return "-" if v is None else f.format(v)


def compare(a):
    """the reference's own comparison (run_performance.sh: LIA lines beside the CPU-only lines of the same shape): per shape the
    hand-picked LIA flags on this box against policies 1 / 1 on its 16 host cores"""
    def load(name):
        p = os.path.join(ROOT, a.dir, f"{a.prefix}_{name}.json")
        return json.load(open(p)) if os.path.exists(p) else None
    names = [n for n, _ in run_matrix.lines()]
    print("| shape (model_in_new_batch) | LIA line: flags | prefill ms | decode tokens/s | CPU-only line (1/1, gpu% 0): prefill ms | decode tokens/s | LIA / CPU-only (prefill time, decode rate) |")
    print("|---|---|---|---|---|---|---|")
    for n in names:
        if not n.startswith("ipex"):
            continue
        kind, rest = n.split("_", 1)
        shape = "_".join(rest.split("_")[:4])
        twin = next((m for m in names if m.startswith(kind.replace("ipex", "") + "_" + shape + "_")), None)
        dc, dl = load(n), load(twin) if twin else None
        rc, rl = (dc or {}).get("result") or {}, (dl or {}).get("result") or {}
        cpu = (f"{rc['prefill_ms']:.0f} | {rc['decode_tokens_per_s']:.2f}" if rc else f"{(dc or {}).get('status', 'not run')} | -")
        lia = (f"{rl['prefill_ms']:.0f} | {rl['decode_tokens_per_s']:.2f}" if rl else f"{str((dl or {}).get('status', 'not run'))[:40]} | -")
        ratio = (f"{rc['prefill_ms'] / rl['prefill_ms']:.1f} x faster, {rl['decode_tokens_per_s'] / rc['decode_tokens_per_s']:.2f} x" if rc and rl else "-")
        print(f"| {shape} | {twin.split(shape + '_')[1] if twin else '-'} | {lia} | {cpu} | {ratio} |")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dir", default="results")
    ap.add_argument("--prefix", default="r06_matrix")
    ap.add_argument("--compare", action="store_true", help="the LIA lines beside the CPU-only lines of the same shape")
    a = ap.parse_args()
    if a.compare:
        return compare(a)
    print("| line (script_model_in_new_batch_policies_gpu%) | status | prefill ms | decode tokens/s | weight stream GB/s over wall (of 63) | dominant in decode (share of wall) | "
          "GEMM rates of the profiled iteration | host GiB of the run | planner's pick: gpu% / policies -> predicted tokens/s | measured with --auto-plan |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for name, _ in run_matrix.lines():
        p = os.path.join(ROOT, a.dir, f"{a.prefix}_{name}.json")
        if not os.path.exists(p):
            print(f"| {name} | not run | | | | | | | | |")
            continue
        d = json.load(open(p))
        r, ws = d.get("result") or {}, d.get("weight_stream") or {}
        pr = d.get("profile_of_warmup_iteration") or {}
        dom = pr.get("dominant_decode") or pr.get("dominant_prefill") or {}
        rates = "; ".join(f"{row[0]} {row[1].split(' (')[0]}: {row[4].strip() or row[5].strip()}" for row in pr.get("rows", []) if row[1].startswith("GEMM"))
        pk = d.get("planner_pick") or {}
        pick = (f"{pk.get('gpu_percentage')}% / {pk.get('prefill_policy')}/{pk.get('decoding_policy')} -> {pk.get('predicted_decode_tokens_per_s')}"
                if "gpu_percentage" in pk else (pk.get("error") or pk.get("note") or "-"))
        ap_p = os.path.join(ROOT, a.dir, f"{a.prefix}_{name}_autoplan.json")
        auto = "-"
        if os.path.exists(ap_p):
            da = json.load(open(ap_p))
            ra, pa = da.get("result") or {}, da.get("auto_plan") or {}
            cl = pa.get("cpu_layers")
            auto = (f"gpu% {pa.get('gpu_percentage')}, {pa.get('prefill_policy')}/{pa.get('decoding_policy')}" + (f", host layers online from {pa.get('cpu_layers_start')}" if cl == -1 else "") +
                    f": prefill {fmt(ra.get('prefill_ms'))} ms, **{fmt(ra.get('decode_tokens_per_s'), '{:.2f}')}** tokens/s") if ra else f"{da.get('status')}: {str(da.get('reason'))[:120]}"
        status = d.get("status", "?")
        if status != "ok":
            import re as _re
            why = str(d.get("reason"))
            m = _re.search(r"needs ([0-9.]+) GiB of host memory but only ([0-9.]+) GiB", why)
            if status.startswith("not run"):
                est_s = d.get("estimated_s")
                status = f"not run: the host-side prefill / decode alone is ~{est_s} s on this box's 16 cores (rates of the lines that ran)"
            else:
                status = (f"refused: {why.split(':')[0]} needs {m.group(1)} GiB of host memory, {m.group(2)} GiB left in the 300 GiB container" if m
                          else f"{status}: {why[:140]}")
        hm = d.get("host_memory") or {}
        print(f"| {name} | {status} | {fmt(r.get('prefill_ms'))} | {fmt(r.get('decode_tokens_per_s'), '{:.2f}')} | "
              f"{fmt(ws.get('gbs_over_wall'), '{:.1f}')} ({fmt(ws.get('fraction_of_63_gbs_link_over_wall'), '{:.2f}')}) | "
              f"{dom.get('what', '-')} ({fmt(dom.get('share_of_wall'), '{:.2f}')}) | {rates or '-'} | {fmt(hm.get('this_run_gib', hm.get('cgroup_peak_gib')))} | {pick} | {auto} |")


if __name__ == "__main__":
    main()
