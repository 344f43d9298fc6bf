"""-m gpu: randomised shape sweep of lia_linear over every GEMM regime (tools/gemm_fuzz.py) against a plain PyTorch fp32
matmul of the same bf16 operands with the reference's rounding points."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [3, 4])
def test_linear_fuzz(seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gemm_fuzz.py"), "120", str(seed)], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "120 cases ok" in r.stdout


def test_attention_fuzz():
    """lia_attention (prefill with the causal mask, decode over a cache, d = 64 / 128, ragged lengths) against a plain PyTorch
    restatement of attentions.py:443-536 with its rounding points (tools/attn_fuzz.py)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "attn_fuzz.py"), "100", "7"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "100 cases ok" in r.stdout
