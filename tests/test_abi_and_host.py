"""CPU-only checks (no GPU, no compute kernels): the C-ABI library loads and exports every symbol
include/lia_hip.h declares; the host-side pieces of the product (policy-2 host attention, TPP layout
converter, NUMA/CXL allocator, packed-layer layout, host facts) behave like the reference's."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "lia_hip.h")


@pytest.fixture(scope="module")
def native():
    from lia_amd import _native
    if not os.path.exists(_native.LIB_PATH):
        subprocess.run(["make", "-C", os.path.join(ROOT, "isca-2025-lia_amd", "csrc"), "-j4"], check=True)
    return _native


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b([a-z_][a-z0-9_]*)\s*\([^;{]*\)\s*;", src)
    return sorted(set(n for n in names if n not in ("defined",)))


def test_header_symbols_are_exported_and_bound(native):
    decl = declared_functions()
    assert len(decl) > 40
    out = subprocess.run(["nm", "-D", "--defined-only", native.LIB_PATH], check=True, capture_output=True, text=True).stdout
    exported = {line.split()[-1] for line in out.splitlines() if line.strip()}
    missing = [n for n in decl if n not in exported]
    assert not missing, f"declared in lia_hip.h but not exported: {missing}"
    unbound = [n for n in decl if n not in native.SIGNATURES]
    assert not unbound, f"declared in lia_hip.h but not bound in _native.SIGNATURES: {unbound}"
    stale = [n for n in native.SIGNATURES if n not in decl]
    assert not stale, f"bound but not declared: {stale}"


def test_library_loads_without_gpu_and_reports_errors(native):
    L = native.lib()
    assert b"gfx950" in L.lia_version()
    h = ctypes.c_void_p()
    rc = L.lia_ctx_create(0, 0, ctypes.byref(h))
    import torch
    if not torch.cuda.is_available():
        assert rc != 0 and len(L.lia_last_error()) > 0   # fails loudly, no CPU fallback
        with pytest.raises((RuntimeError, ValueError, MemoryError)):
            native.check(rc, "lia_ctx_create")
    else:
        L.lia_ctx_destroy(h)


def test_pack_offsets_layout(native):
    from lia_amd import ops
    d = ops.make_desc(7168, 56, 28672)
    offs, total = ops.pack_offsets(d)
    H, F = 7168, 28672
    assert all(o % 256 == 0 for i, o in enumerate(offs) if i not in (4, 6, 5, 7))
    assert offs[4] == offs[2] + 2 * H * H and offs[6] == offs[4] + 2 * H * H      # q|k|v weights adjacent
    assert offs[5] == offs[3] + 2 * H and offs[7] == offs[5] + 2 * H              # q|k|v biases adjacent
    params = 2 * (4 * H * H + 2 * H * F + 9 * H + F)
    assert params <= total <= params + 16 * 256
    assert params == 1233311744                                                    # bytes/layer
    with pytest.raises(ValueError):
        ops.pack_offsets(ops.make_desc(250, 5, 1024))
    assert ops.workspace_bytes(d, 64) > 0
    # the workspace size is monotonic in the row count: a context sized for the largest call of a forward (e.g. 292 rows of a
    # policy-0 decode slab) must also serve its smaller calls (a 256-row prefill wants split-K slabs, r01 did not reserve them)
    sizes = [ops.workspace_bytes(ops.make_desc(2048, 32, 8192), r) for r in (1, 64, 255, 256, 257, 292, 1024, 16384)]
    assert sizes == sorted(sizes) and len(set(sizes)) == len(sizes)


@pytest.mark.parametrize("B,T,pos0,heads,d", [(2, 1, 8, 4, 32), (3, 1, 33, 4, 64), (2, 1, 40, 4, 128), (2, 5, 0, 8, 64),
                                              (1, 1, 0, 2, 128)])
def test_host_attention_matches_oracle(native, oracle, B, T, pos0, heads, d):
    """lia_host_attention (product, AVX-512) == the oracle's restatement of Krnl.cpp:513-842, incl. the in-place
    append of the new K/V rows and a minibatch offset into a wider cache."""
    L = native.lib()
    H = heads * d
    rs = np.random.RandomState(7)
    q, k, v = (synth.f32_to_bf16_bits(rs.standard_normal((B, T, H)).astype(np.float32)) for _ in range(3))
    Bc, b0, smax = B + 2, 1, pos0 + T + 1
    kc = synth.f32_to_bf16_bits(rs.standard_normal((smax, Bc, heads, d)).astype(np.float32))
    vc = synth.f32_to_bf16_bits(rs.standard_normal((smax, Bc, heads, d)).astype(np.float32))
    kc_o, vc_o = np.ascontiguousarray(kc[:, b0:b0 + B]), np.ascontiguousarray(vc[:, b0:b0 + B])
    oracle.lib().lia_oracle_kv_store(k.ctypes.data, kc_o.ctypes.data, B, T, H, pos0)
    oracle.lib().lia_oracle_kv_store(v.ctypes.data, vc_o.ctypes.data, B, T, H, pos0)
    ref = oracle.attention(q, kc_o, vc_o, pos0 + T, heads, policy_gpu=False)
    out = np.zeros_like(q)
    rc = L.lia_host_attention(q.ctypes.data, k.ctypes.data, v.ctypes.data, kc.ctypes.data, vc.ctypes.data, out.ctypes.data,
                              B, T, pos0, heads, d, Bc, b0, 3)
    assert rc == 0
    err = np.abs(synth.bf16_bits_to_f32(out) - synth.bf16_bits_to_f32(ref))
    assert err.max() <= 0.02 and (out == ref).mean() > 0.97
    assert (kc[pos0:pos0 + T, b0:b0 + B].reshape(T, B, H) == k.transpose(1, 0, 2)).all()      # rows appended in place
    assert (kc[:, 0] == kc[:, 0]).all() and (vc[pos0:pos0 + T, b0:b0 + B].reshape(T, B, H) == v.transpose(1, 0, 2)).all()
    # argument errors come back as codes, not crashes
    assert L.lia_host_attention(None, k.ctypes.data, v.ctypes.data, kc.ctypes.data, vc.ctypes.data, out.ctypes.data, B, T, pos0,
                                heads, d, Bc, b0, 1) == native.LIA_ERR_MISSING
    assert L.lia_host_attention(q.ctypes.data, k.ctypes.data, v.ctypes.data, kc.ctypes.data, vc.ctypes.data, out.ctypes.data, B, T,
                                pos0, heads, d, Bc, Bc, 1) == native.LIA_ERR_INVALID


def test_tpp_layout_converter(native, oracle):
    L = native.lib()
    N, K = 64, 192
    w = np.arange(N * K, dtype=np.uint16).reshape(N, K)
    blocked = np.zeros(N * K, np.uint16)
    assert L.lia_tpp_block(w.ctypes.data, blocked.ctypes.data, N, K) == 0
    assert (blocked.reshape(N // 16, K // 64, 32, 16, 2) == oracle.tpp_block(w)).all()
    back = np.zeros_like(w)
    assert L.lia_tpp_unblock(blocked.ctypes.data, back.ctypes.data, N, K) == 0
    assert (back == w).all()
    assert L.lia_tpp_unblock(blocked.ctypes.data, back.ctypes.data, 60, K) == native.LIA_ERR_INVALID   # tpp_fallback shapes


def test_numa_shim_matches_reference_build(native):
    """Same four exports and behaviour as the reference's lia/cxl/numa_alloc.c (compiled from the reference's own
    source into oracle/_ref when the reference tree is present)."""
    L = native.lib()
    if not L.lia_numa_available():
        pytest.skip("no NUMA support on this host")
    size = 1 << 20
    p = L.numa_alloc_node(size, 0)
    assert p
    buf = (ctypes.c_uint8 * size).from_address(p)
    buf[0], buf[size - 1] = 7, 9
    assert buf[0] == 7 and buf[size - 1] == 9
    L.numa_free_node(p, size)
    assert not L.numa_alloc_node(size, 4096)          # non-existent node -> NULL (+ stderr), like the reference
    nodes = (ctypes.c_int * 1)(0)
    assert L.lia_numa_set_interleave_nodes(nodes, 1) == 0
    p = L.numa_alloc_interleave(size)
    assert p
    L.numa_free_node(p, size)
    bad = (ctypes.c_int * 1)(4096)
    assert L.lia_numa_set_interleave_nodes(bad, 1) == native.LIA_ERR_INVALID
    ref_path = os.path.join(ROOT, "oracle", "_ref", "libnuma_alloc_ref.so")
    if os.path.exists(ref_path):
        R = ctypes.CDLL(ref_path)
        R.numa_alloc_node.restype = ctypes.c_void_p
        R.numa_alloc_node.argtypes = [ctypes.c_size_t, ctypes.c_int]
        R.numa_free_node.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
        rp = R.numa_alloc_node(size, 0)
        assert rp                                        # reference allocates on node 0 as ours does
        R.numa_free_node(rp, size)
        R.numa_alloc_interleave.restype = ctypes.c_void_p
        R.numa_alloc_interleave.argtypes = [ctypes.c_size_t]
    

def test_hostinfo():
    from lia_amd import hostinfo
    n = hostinfo.usable_cpus()
    assert 1 <= n <= (os.cpu_count() or 1)
    assert 1 <= hostinfo.default_host_threads(1) <= n
    assert hostinfo.default_host_threads(8) >= 1
    assert isinstance(hostinfo.cpu_model(), str) and "avx512f" in hostinfo.isa_flags()


def test_pin_node_follows_the_switch(monkeypatch):
    """LIA_PIN_NODE: the NUMA node the host threads are confined to (-1 = off); unset = the GPU's node from sysfs"""
    from lia_amd import hostinfo
    monkeypatch.setattr(hostinfo, "gpu_numa_node", lambda dev=0: 5)
    monkeypatch.delenv("LIA_PIN_NODE", raising=False)
    assert hostinfo.pin_node(0) == 5
    monkeypatch.setenv("LIA_PIN_NODE", "-1")
    assert hostinfo.pin_node(0) == -1
    monkeypatch.setenv("LIA_PIN_NODE", "1")
    assert hostinfo.pin_node(3) == 1
    monkeypatch.setenv("LIA_PIN_NODE", "")
    assert hostinfo.pin_node(0) == 5
    assert hostinfo.pin_to_node(10 ** 6) == 0                  # no such node: nothing pinned


def test_host_team_is_the_attention_teams_count():
    """the whole-layer host team: a fixed count (r03's throttle governor and its two switches are gone)"""
    from lia_amd import hostinfo
    assert hostinfo.HostTeam(16).threads == 16 and hostinfo.HostTeam(0).threads == 1 and not hasattr(hostinfo, "HostTeamGovernor")


def test_shapes_and_flag_defaults():
    from lia_amd.model import SHAPES, resolve_shape
    s = resolve_shape("facebook/opt-30b")
    assert (s.hidden, s.heads, s.ffn, s.layers, s.head_dim) == (7168, 56, 28672, 48, 128)
    assert s.layer_param_bytes() == 1233311744
    assert resolve_shape("opt-175b").layer_param_bytes() == 2 * (4 * 12288 ** 2 + 2 * 12288 * 49152 + 9 * 12288 + 49152)
    assert int(48 * 10 / 100) == 4                       # n_gpu_layers at gpu%=10 (lia/modeling_opt.py:1182)
    with pytest.raises(ValueError):
        resolve_shape("opt-350m")                         # post-LN variant: out of scope
    assert set(SHAPES) >= {"opt-125m", "opt-30b", "opt-66b", "opt-175b"}


def test_checkpoint_state_dict_conversion(native, oracle):
    """HF-named state dict (one linear stored TPP-blocked, as an IPEX-prepacked checkpoint would) -> packed layout."""
    import torch
    from lia_amd.checkpoint import state_dict_to_numpy
    vocab, max_pos, H, F, L = 64, 16, 128, 256, 2
    m = synth.make_model(5, vocab, max_pos, H, F, L)

    def t(bits):
        return torch.from_numpy(np.ascontiguousarray(bits).view(np.int16)).view(torch.bfloat16)

    hf = {"ln1": "self_attn_layer_norm", "q": "self_attn.q_proj", "k": "self_attn.k_proj", "v": "self_attn.v_proj",
          "out": "self_attn.out_proj", "ln2": "final_layer_norm", "fc1": "fc1", "fc2": "fc2"}
    sd = {"model.decoder.embed_tokens.weight": t(m["embed_tokens"]), "model.decoder.embed_positions.weight": t(m["embed_positions"]),
          "model.decoder.final_layer_norm.weight": t(m["final_ln_w"]), "model.decoder.final_layer_norm.bias": t(m["final_ln_b"])}
    for i, lw in enumerate(m["layers"]):
        for n, v in lw.items():
            base, kind = n.rsplit("_", 1)
            sd[f"model.decoder.layers.{i}.{hf[base]}.{'weight' if kind == 'w' else 'bias'}"] = t(v)
    sd["model.decoder.layers.1.fc1.weight"] = t(oracle.tpp_block(m["layers"][1]["fc1_w"]))     # blocked on disk
    got = state_dict_to_numpy(sd, dict(hidden_size=H, ffn_dim=F, num_hidden_layers=L))
    assert (got["embed_tokens"] == m["embed_tokens"]).all() and (got["final_ln_b"] == m["final_ln_b"]).all()
    for a, b in zip(got["layers"], m["layers"]):
        for n in synth.LAYER_TENSORS:
            assert a[n].shape == b[n].shape and (a[n] == b[n]).all(), n


@pytest.mark.parametrize("B,T,pos0", [(2, 6, 0), (3, 1, 7)])
def test_host_layer_forward_matches_oracle_policy1(native, oracle, B, T, pos0):
    """lia_host_layer_forward (product, AVX-512-BF16) vs the oracle's policy-1 restatement (fused-bias linears,
    fp32 attention).  vdpbf16ps sums bf16 pairs before accumulating, so agreement is to rounding, not bitwise."""
    from lia_amd import ops
    L = native.lib()
    H, heads, F = 256, 4, 1024
    W = synth.make_layer(3, H, F, 0.08)
    x = synth.make_hidden(4, B, T, H)
    d = H // heads
    smax = pos0 + T + 2
    rs = np.random.RandomState(9)
    kc = synth.f32_to_bf16_bits(rs.standard_normal((smax, B, heads, d)).astype(np.float32))
    vc = synth.f32_to_bf16_bits(rs.standard_normal((smax, B, heads, d)).astype(np.float32))
    kc_o, vc_o = kc.copy(), vc.copy()
    ref = oracle.layer_forward(1, W, x, kc_o, vc_o, pos0, heads)
    desc = ops.make_desc(H, heads, F)
    ws = [np.ascontiguousarray(W[n]) for n in synth.LAYER_TENSORS]
    arr = (ctypes.c_void_p * 16)(*[w.ctypes.data for w in ws])
    y = np.zeros_like(x)
    rc = L.lia_host_layer_forward(ctypes.byref(desc), ctypes.byref(arr), x.ctypes.data, y.ctypes.data, kc.ctypes.data, vc.ctypes.data,
                                  smax, B, B, T, pos0, 0, 4)
    assert rc == 0, L.lia_last_error()
    err = np.abs(synth.bf16_bits_to_f32(y) - synth.bf16_bits_to_f32(ref))
    assert err.max() <= 0.07 and (y == ref).mean() > 0.9, (err.max(), (y == ref).mean())
    assert (kc[pos0:pos0 + T] == kc_o[pos0:pos0 + T]).mean() > 0.97
    bad = (ctypes.c_void_p * 16)(*[None] * 16)
    assert L.lia_host_layer_forward(ctypes.byref(desc), ctypes.byref(bad), x.ctypes.data, y.ctypes.data, kc.ctypes.data, vc.ctypes.data,
                                    smax, B, B, T, pos0, 0, 1) == native.LIA_ERR_MISSING


@pytest.mark.parametrize("n_layers", [1, 2, 3])
def test_host_layers_forward_equals_layer_by_layer(native, n_layers):
    """lia_host_layers_forward (the decode step of policy 1 over consecutive layers in ONE OpenMP region, hidden state ping-ponging
    between two buffers) == lia_host_layer_forward called once per layer: hidden state and every cache bit for bit; the return
    value names the buffer that holds the result; prefill-sized steps (B * T > 256) are refused."""
    from lia_amd import ops
    L = native.lib()
    if not L.lia_host_has_avx512_bf16():
        pytest.skip("host without AVX-512-BF16")
    H, heads, F, B, T, pos0 = 256, 4, 1024, 3, 1, 5
    d = H // heads
    smax = pos0 + T + 1
    desc = ops.make_desc(H, heads, F)
    Ws = [synth.make_layer(30 + i, H, F, 0.08) for i in range(n_layers)]
    flat = [[np.ascontiguousarray(W[n]) for n in synth.LAYER_TENSORS] for W in Ws]
    rs = np.random.RandomState(11)
    caches = [(synth.f32_to_bf16_bits(rs.standard_normal((smax, B, heads, d)).astype(np.float32)),
               synth.f32_to_bf16_bits(rs.standard_normal((smax, B, heads, d)).astype(np.float32))) for _ in range(n_layers)]
    x0 = synth.make_hidden(12, B, T, H)
    # reference: one call per layer
    ref_c = [(k.copy(), v.copy()) for k, v in caches]
    a, b = x0.copy(), np.zeros_like(x0)
    for i in range(n_layers):
        arr = (ctypes.c_void_p * 16)(*[w.ctypes.data for w in flat[i]])
        assert L.lia_host_layer_forward(ctypes.byref(desc), ctypes.byref(arr), a.ctypes.data, b.ctypes.data, ref_c[i][0].ctypes.data,
                                        ref_c[i][1].ctypes.data, smax, B, B, T, pos0, 0, 3) == 0, L.lia_last_error()
        a, b = b, a
    wt = (ctypes.c_void_p * (16 * n_layers))(*[w.ctypes.data for fl in flat for w in fl])
    kt = (ctypes.c_void_p * n_layers)(*[k.ctypes.data for k, _ in caches])
    vt = (ctypes.c_void_p * n_layers)(*[v.ctypes.data for _, v in caches])
    x, y = x0.copy(), np.zeros_like(x0)
    where = L.lia_host_layers_forward(ctypes.byref(desc), n_layers, wt, x.ctypes.data, y.ctypes.data, kt, vt, smax, B, B, T, pos0, 0, 3)
    assert where == n_layers % 2, (where, L.lia_last_error())
    got = y if where == 1 else x
    assert (got == a).all()
    for (k, v), (rk, rv) in zip(caches, ref_c):
        assert (k == rk).all() and (v == rv).all()
    big = np.zeros((2, 200, H), np.uint16)          # B * T = 400 > 256: not a decode step
    assert L.lia_host_layers_forward(ctypes.byref(desc), n_layers, wt, big.ctypes.data, big.copy().ctypes.data, kt, vt, 512, 2, 2, 200, 0, 0, 1) == native.LIA_ERR_INVALID
    assert L.lia_host_layers_forward(ctypes.byref(desc), n_layers, None, x.ctypes.data, y.ctypes.data, kt, vt, smax, B, B, T, pos0, 0, 1) == native.LIA_ERR_MISSING


def test_host_kernels_report_scratch_failure_instead_of_crashing(native):
    """a worker thread that cannot get its scratch block (refused here by lia_host_thread_scratch_limit; the same path as a failed
    aligned_alloc): the call returns LIA_ERR_MEMORY with a message, the team leaves its region in step, and the next call with
    the limit lifted computes as before; F <= 0 is a shape error"""
    from lia_amd import ops
    L = native.lib()
    if not L.lia_host_has_avx512_bf16():
        pytest.skip("host without AVX-512-BF16")
    H, heads, F, B, T, pos0 = 256, 4, 1024, 3, 1, 5
    d = H // heads
    smax = pos0 + T + 1
    desc = ops.make_desc(H, heads, F)
    W = synth.make_layer(30, H, F, 0.08)
    flat = [np.ascontiguousarray(W[n]) for n in synth.LAYER_TENSORS]
    arr = (ctypes.c_void_p * 16)(*[w.ctypes.data for w in flat])
    rs = np.random.RandomState(11)
    k0 = synth.f32_to_bf16_bits(rs.standard_normal((smax, B, heads, d)).astype(np.float32))
    v0 = synth.f32_to_bf16_bits(rs.standard_normal((smax, B, heads, d)).astype(np.float32))
    x = synth.make_hidden(12, B, T, H)

    def layer(limit):
        L.lia_host_thread_scratch_limit(limit)
        try:
            k, v, y = k0.copy(), v0.copy(), np.zeros_like(x)
            rc = L.lia_host_layer_forward(ctypes.byref(desc), ctypes.byref(arr), x.ctypes.data, y.ctypes.data, k.ctypes.data, v.ctypes.data,
                                          smax, B, B, T, pos0, 0, 3)
            return rc, y
        finally:
            L.lia_host_thread_scratch_limit(0)

    rc0, y0 = layer(0)
    assert rc0 == 0
    rc1, _ = layer(64)                                       # 64 bytes per thread: every linear tile and score row is refused
    assert rc1 == native.LIA_ERR_MEMORY and b"scratch" in L.lia_last_error()
    rc2, y2 = layer(0)
    assert rc2 == 0 and (y2 == y0).all()                     # the flag does not leak into the next call
    L.lia_host_thread_scratch_limit(64)
    try:
        xs = np.zeros((4, 64), np.uint16)
        ws = np.zeros((32, 64), np.uint16)
        ys = np.zeros((4, 32), np.uint16)
        assert L.lia_host_linear(xs.ctypes.data, ws.ctypes.data, None, None, ys.ctypes.data, 4, 32, 64, 0, 2) == native.LIA_ERR_MEMORY
        wt = (ctypes.c_void_p * 16)(*[w.ctypes.data for w in flat])
        kt, vt = (ctypes.c_void_p * 1)(k0.ctypes.data), (ctypes.c_void_p * 1)(v0.ctypes.data)
        assert L.lia_host_layers_forward(ctypes.byref(desc), 1, wt, x.copy().ctypes.data, np.zeros_like(x).ctypes.data, kt, vt, smax, B, B, T,
                                         pos0, 0, 3) == native.LIA_ERR_MEMORY
    finally:
        L.lia_host_thread_scratch_limit(0)
    bad = ops.make_desc(H, heads, 0)
    assert L.lia_host_layer_forward(ctypes.byref(bad), ctypes.byref(arr), x.ctypes.data, np.zeros_like(x).ctypes.data, k0.copy().ctypes.data,
                                    v0.copy().ctypes.data, smax, B, B, T, pos0, 0, 3) == native.LIA_ERR_INVALID


_HOST_LINEAR_CASES = [(64, 1000, 2112, 1, 1, 0), (7, 77, 96, 1, 1, 1), (1, 50, 4096, 0, 0, 0), (130, 130, 320, 1, 0, 1), (64, 1536, 1024, 0, 1, 1),
                      (5, 6, 64, 0, 1, 0), (300, 40, 64, 1, 1, 1)]


def _host_linear_reference(x, w, b, r, relu):
    """the CPU policy's linear in exact arithmetic + its two roundings (tpp_linear_bias: bias joins the fp32 accumulator before
    the single rounding; `+ residual` is a second bf16 op) -- TPPGEMMKrnl.h:89-176 / decoder.py via DESIGN.md section 3"""
    acc = synth.bf16_bits_to_f32(x).astype(np.float64) @ synth.bf16_bits_to_f32(w).astype(np.float64).T
    if b is not None:
        acc = acc + synth.bf16_bits_to_f32(b).astype(np.float64)
    t = synth.bf16_bits_to_f32(synth.f32_to_bf16_bits(acc.astype(np.float32)))
    if relu:
        t = np.where(t < 0, np.float32(0), t)
    if r is not None:
        t = synth.bf16_bits_to_f32(synth.f32_to_bf16_bits(synth.bf16_bits_to_f32(r) + t))
    return synth.f32_to_bf16_bits(t)


def test_host_linear_edge_shapes_and_epilogue(native, tmp_path):
    """lia_host_linear (4 x 6 register blocks in the decode kernel): N not a multiple of the block or of 16, M not a multiple of
    4, K-chunk tails, M > 256 (the generic kernel), bias / ReLU / residual in every combination that the layer uses.  Against
    exact arithmetic: <= 1 % of the outputs may differ by one bf16 ulp (fp32 summation order), none by more.  (A child process:
    the reference values are computed in float64 numpy next to an OpenMP team.)"""
    import subprocess
    import sys
    if not native.lib().lia_host_has_avx512_bf16():
        pytest.skip("host without AVX-512-BF16")
    code = f"""
import sys
sys.path[:0] = {[os.path.dirname(os.path.abspath(__file__)), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "isca-2025-lia_amd")]!r}
import numpy as np, synth
import test_abi_and_host as T
from lia_amd import _native as N
L = N.lib()
rs = np.random.RandomState(1)
for (M, n, k, relu, use_b, use_r) in T._HOST_LINEAR_CASES:
    x = synth.f32_to_bf16_bits(rs.standard_normal((M, k)).astype(np.float32)); w = synth.f32_to_bf16_bits((0.02 * rs.standard_normal((n, k))).astype(np.float32))
    b = synth.f32_to_bf16_bits((0.1 * rs.standard_normal(n)).astype(np.float32)) if use_b else None
    r = synth.f32_to_bf16_bits(rs.standard_normal((M, n)).astype(np.float32)) if use_r else None
    y = np.full((M + 1, n), 0x7fc1, np.uint16)                       # a guard row behind the output: must stay untouched
    rc = L.lia_host_linear(x.ctypes.data, w.ctypes.data, b.ctypes.data if use_b else None, r.ctypes.data if use_r else None, y.ctypes.data, M, n, k, relu, 3)
    assert rc == 0, L.lia_last_error()
    assert (y[M] == 0x7fc1).all(), "wrote past the output"
    ref = T._host_linear_reference(x, w, b, r, relu)
    neq = (ref != y[:M])
    ulp = np.abs(ref.astype(np.int32) - y[:M].astype(np.int32))
    assert neq.mean() <= 0.01 and ulp.max() <= 1, (M, n, k, float(neq.mean()), int(ulp.max()))
print("ok")
"""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout[-500:], r.stderr[-2000:])


def test_numa_alloc_tensor_wrappers(native):
    """numa_alloc_tensor / numa_free_tensor keep the reference's contract (lia/cxl/numa_alloc.py:28-55)."""
    import torch
    from lia_amd.cxl import numa_alloc
    if not native.lib().lia_numa_available():
        pytest.skip("no NUMA support on this host")
    numa_alloc.set_cxl_nodes([0])
    t = numa_alloc.numa_alloc_tensor((4, 8, 16), torch.bfloat16)
    assert t is not None and t.shape == (4, 8, 16) and t.dtype == torch.bfloat16
    t.fill_(1.5)
    assert float(t.float().sum()) == 1.5 * 4 * 8 * 16
    numa_alloc.numa_free_tensor(t)
    with pytest.raises(ValueError):
        numa_alloc.set_cxl_nodes([999])


def test_host_allocation_guard(monkeypatch):
    """An allocation plan beyond the container's memory budget is refused up front (MemoryError), not attempted."""
    from lia_amd import hostinfo
    b = hostinfo.host_memory_budget()
    assert b is None or b > 0
    monkeypatch.setattr(hostinfo, "host_memory_budget", lambda: 100 << 30)
    hostinfo.check_host_allocation(50 << 30, "fits")
    with pytest.raises(MemoryError):
        hostinfo.check_host_allocation(90 << 30, "opt-175b streamed layers")


def test_host_linear_restates_the_references_tpp_linear_tests(native):
    """tests/cpu/test_tpp_linear.py:104-305 on this path's policy-1 linear: x = rand(1, 4, 4096), nn.Linear(4096, 4096) in bf16,
    with bias (:124-150 `tpp_linear_bias`), without, + ReLU (:229-250 `tpp_linear_relu`, the fc1 form), + the input as residual
    (:269-290 `tpp_linear_add`, the out_proj / fc2 form); the reference asserts equality with eager bf16 at its TestCase's bf16
    precision.  Here: against torch's eager bf16 module, at most one bf16 ulp apart (1e-4 absolute around zero, where the fp32
    summation order shows), >= 99 % bit-identical.  (A child process: torch's
    OpenMP runtime next to the library's team.)"""
    import subprocess
    import sys
    if not native.lib().lia_host_has_avx512_bf16():
        pytest.skip("host without AVX-512-BF16")
    code = f"""
import sys
sys.path[:0] = {[os.path.join(ROOT, "isca-2025-lia_amd")]!r}
import numpy as np, torch
from lia_amd import _native as N
L = N.lib()
torch.manual_seed(128)
torch.set_num_threads(2)
bits = lambda t: t.contiguous().view(torch.int16).numpy().view(np.uint16)
for name, bias, relu, add in [("linear", True, 0, False), ("linear_nobias", False, 0, False), ("linear_relu", False, 1, False),
                              ("linear_relu_bias", True, 1, False), ("linear_add", False, 0, True), ("linear_add_bias", True, 0, True)]:
    x = torch.rand(1, 4, 4096).to(torch.bfloat16)
    mlp = torch.nn.Linear(4096, 4096, bias=bias).eval().to(torch.bfloat16)
    with torch.no_grad():
        ref = mlp(x)
        if relu: ref = torch.nn.functional.relu(ref)
        if add: ref = ref + x
    xb, wb = bits(x[0]), bits(mlp.weight.detach())
    bb = bits(mlp.bias.detach()) if bias else None
    y = np.zeros((4, 4096), np.uint16)
    rc = L.lia_host_linear(xb.ctypes.data, wb.ctypes.data, bb.ctypes.data if bias else None, xb.ctypes.data if add else None, y.ctypes.data, 4, 4096, 4096, relu, 4)
    assert rc == 0, L.lia_last_error()
    r = bits(ref[0])
    rf, yf = ref[0].float().numpy(), torch.from_numpy(y.view(np.int16)).view(torch.bfloat16).float().numpy()
    quantum = np.exp2(np.floor(np.log2(np.maximum(np.abs(rf), 1e-30))) - 7)       # one bf16 ulp at the reference value
    err = np.abs(rf - yf)
    same = float((r == y).mean())
    assert (err <= np.maximum(quantum, 1e-4)).all() and same >= 0.99, (name, float(err.max()), same)
    print(name, "identical", same)
print("ok")
"""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout[-800:], r.stderr[-2000:])


def test_pack10_validate_refuses_garbage_on_the_host():
    """lia_pack10_validate is host code (no GPU): a buffer that is not a pack10 header is refused with the failed check's code"""
    import ctypes
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "isca-2025-lia_amd"))
    from lia_amd import _native as N
    L = N.lib()
    junk = np.zeros(4096, np.uint8)
    assert L.lia_pack10_validate(junk.ctypes.data, 4096, 1024) == -2        # no magic
    assert L.lia_pack10_validate(junk.ctypes.data, 100, 1024) == -1         # shorter than a header
    assert L.lia_pack10_validate(None, 4096, 1024) == -1
    hdr = np.zeros(64, np.uint32)
    hdr[0], hdr[1] = 0x3031504c, 2                                           # 'LP10', version 2, but n = 0
    assert L.lia_pack10_validate(hdr.ctypes.data, 256, 1024) == -3


def test_packed_directory_in_a_removed_wire_format_is_refused_by_name(tmp_path):
    """a directory written with --wire pack11 / pack12 by an earlier build ended in a bare KeyError"""
    import json
    sys.path.insert(0, os.path.join(ROOT, "isca-2025-lia_amd"))
    from lia_amd import scheduler
    with pytest.raises(ValueError, match="pack10"):
        scheduler.wire_format_code(12)
    with pytest.raises(ValueError, match="pack10"):
        scheduler.wire_format_code("pack11")
    assert [scheduler.wire_format_code(f) for f in ("raw", "pack10", 0, 10, False, True)] == [0, 10, 0, 10, 0, 10]
