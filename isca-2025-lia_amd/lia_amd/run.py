"""Launcher: counterpart of llm/run.py (reference :27-216 flag surface, :234-287 hand-off to the harness).
The reference re-launches single_instance/run_generation.py in a subprocess; here the harness is called
in-process with the same flags.

    python -m lia_amd.run --benchmark -m facebook/opt-30b --dtype bfloat16 --ipex --input-tokens 256 \
        --max-new-tokens 32 --batch-size 64 --token-latency --num-iter 10 --num-warmup 2 --greedy \
        --prefill-policy 0 --decoding-policy 2 --gpu-percentage 10 --num-minibatch 2 --pin-weight
"""
import os
import sys

# park OpenMP teams between host-attention bursts; must precede the first libgomp load (torch import)
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
os.environ.setdefault("GOMP_SPINCOUNT", "0")


def main(argv=None):
    from . import run_generation
    return run_generation.main(argv)


if __name__ == "__main__":
    main(sys.argv[1:])
