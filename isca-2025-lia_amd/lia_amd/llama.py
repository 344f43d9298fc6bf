"""Llama-family model + scheduler (BASELINE.json config 4: Llama-3-8B, gpu%=100, bs=128, in 1024 / out 128).

A build-defined extension: in the reference only OPT carries the LIA policies (LlamaDecoderLayer_forward takes no
`policy`, decoder.py:121-169; gpu_percentage=100 never copies activations back, lia/modeling_opt.py:1262-1267 --
SURVEY.md quirk 3).  Semantics chosen here: layers [0, n_gpu) are HBM-resident; the remaining ones stream their packed
weights through the same WeightPipeline; the KV cache of every layer lives in HBM (the host-attention policies are
OPT-only), i.e. every layer runs "policy 3" arithmetic.  Arithmetic = HF transformers' eager bf16 Llama.
"""
import ctypes
from dataclasses import dataclass

import numpy as np
import torch

from . import _native as N
from . import ops
from .model import LayerStore
from .scheduler import WeightPipeline, _KV_SERIAL

LLAMA_TENSORS = ("in_norm_w", "q_w", "k_w", "v_w", "o_w", "post_norm_w", "gate_w", "up_w", "down_w")


@dataclass(frozen=True)
class LlamaShape:
    name: str
    hidden: int
    heads: int
    kv_heads: int
    ffn: int
    layers: int
    vocab: int
    max_pos: int = 8192
    rope_theta: float = 500000.0
    rms_eps: float = 1e-5

    @property
    def head_dim(self):
        return self.hidden // self.heads


LLAMA_SHAPES = {s.name: s for s in (
    LlamaShape("llama-3-8b", 4096, 32, 8, 14336, 32, 128256),
    LlamaShape("llama-2-7b", 4096, 32, 32, 11008, 32, 32000, max_pos=4096, rope_theta=10000.0),
)}


def resolve_llama_shape(name):
    key = name.lower().split("/")[-1].replace("meta-", "")
    if key not in LLAMA_SHAPES:
        raise ValueError(f"unknown Llama shape {name!r}; known: {sorted(LLAMA_SHAPES)}")
    return LLAMA_SHAPES[key]


GU_BLOCK = 32      # lia_hip.h, lia_llama_desc.gu_block


def interleave_gate_up(gate, up, block=GU_BLOCK):
    """[F, H] gate.w and up.w -> the [2F, H] block of the layer buffer: `block` gate rows, the matching `block` up rows, ..."""
    F, H = gate.shape
    assert up.shape == (F, H) and F % block == 0
    return np.stack([gate.reshape(F // block, block, H), up.reshape(F // block, block, H)], axis=1).reshape(2 * F, H)


def rope_tables(max_pos, d, theta):
    """cos/sin [max_pos, d] in bf16, computed like LlamaRotaryEmbedding.forward: fp32 inv_freq, fp32 angles, cast."""
    inv_freq = 1.0 / (theta ** (torch.arange(0, d, 2, dtype=torch.int64).float() / d))
    freqs = torch.outer(torch.arange(max_pos, dtype=torch.float32), inv_freq)
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos().to(torch.bfloat16).cuda(), emb.sin().to(torch.bfloat16).cuda()


class LiaLlamaModel:
    family = "llama"

    def __init__(self, shape):
        self.shape = shape
        # gate.w / up.w live interleaved in the layer buffer, 32 gate rows then the matching 32 up rows (lia_hip.h, gu_block): a
        # 64-column tile of the gate|up projection then holds both factors of 32 outputs and the prefill GEMM's epilogue
        # writes silu(gate) * up itself.  The model is the same function of the same weights; only their order in memory differs.
        self.desc = N.LlamaDesc(shape.hidden, shape.heads, shape.kv_heads, shape.ffn, shape.rms_eps, GU_BLOCK)
        offs = (ctypes.c_size_t * 9)()
        total = ctypes.c_size_t()
        N.check(N.lib().lia_llama_pack_offsets(ctypes.byref(self.desc), ctypes.byref(offs), ctypes.byref(total)), "lia_llama_pack_offsets")
        self.offsets, self.layer_bytes = list(offs), total.value
        self.layers = [LayerStore(self.desc, self.offsets, self.layer_bytes) for _ in range(shape.layers)]
        self.embed_tokens = self.lm_head = self.final_norm_w = None
        self.placed_for = None

    def _pack_numpy(self, st, tensors):
        flat = np.zeros(self.layer_bytes, np.uint8)
        F, H = self.shape.ffn, self.shape.hidden
        gi, ui = LLAMA_TENSORS.index("gate_w"), LLAMA_TENSORS.index("up_w")
        assert self.offsets[ui] == self.offsets[gi] + F * H * 2, "gate.w and up.w must be adjacent in the layer buffer"
        for i, n in enumerate(LLAMA_TENSORS):
            if n == "up_w":
                continue                                   # written together with gate_w
            a = np.ascontiguousarray(tensors[n], dtype=np.uint16)
            if n == "gate_w":
                a = interleave_gate_up(a.reshape(F, H), np.ascontiguousarray(tensors["up_w"], dtype=np.uint16).reshape(F, H))
            a = a.reshape(-1).view(np.uint8)
            flat[self.offsets[i]: self.offsets[i] + a.size] = a
        st._free()
        st._np, st.tier = flat, "pageable"

    @classmethod
    def from_numpy(cls, shape, m):
        self = cls(shape)
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int16)).view(torch.bfloat16).cuda()  # noqa: E731
        self.embed_tokens, self.lm_head, self.final_norm_w = dev(m["embed_tokens"]), dev(m["lm_head"]), dev(m["final_norm_w"])
        for st, lw in zip(self.layers, m["layers"]):
            self._pack_numpy(st, lw)
        return self

    @classmethod
    def random_init(cls, shape, seed=0, n_gpu_layers=None, pin_weight=True, pack=0):
        """pack = 0 / 10: wire format of the pinned streamed layers (lia_pack10.hip), as for the OPT model"""
        self = cls(shape)
        n_gpu = shape.layers if n_gpu_layers is None else n_gpu_layers
        from . import hostinfo
        hostinfo.check_host_allocation((shape.layers - n_gpu) * self.layer_bytes, f"{shape.name}: streamed layers")
        g = torch.Generator(device="cuda")
        g.manual_seed(seed * 100003)
        H, F, KD = shape.hidden, shape.ffn, shape.kv_heads * shape.head_dim
        draw = lambda *s: (0.02 * torch.randn(*s, generator=g, device="cuda", dtype=torch.float32)).to(torch.bfloat16)  # noqa: E731
        self.embed_tokens, self.lm_head = draw(shape.vocab, H), draw(shape.vocab, H)
        self.final_norm_w = torch.ones(H, dtype=torch.bfloat16, device="cuda")
        dims = {"q_w": (H, H), "k_w": (KD, H), "v_w": (KD, H), "o_w": (H, H), "gate_w": (F, H), "up_w": (F, H), "down_w": (H, F)}
        for li, st in enumerate(self.layers):
            g.manual_seed(seed * 100003 + li + 1)
            flat = torch.zeros(self.layer_bytes // 2, dtype=torch.bfloat16, device="cuda")
            for i, n in enumerate(LLAMA_TENSORS):
                o = self.offsets[i] // 2
                if n in dims:
                    r, c = dims[n]
                    flat[o:o + r * c] = draw(r, c).reshape(-1)
                else:
                    flat[o:o + H] = 1.0
            st.set_from_device(flat.view(torch.uint8))
            if li >= n_gpu:
                st.to_pinned(pack)
        torch.cuda.synchronize()
        self.placed_for = (n_gpu, True, False, LayerStore._fmt_of(pack))
        return self

    def place(self, n_gpu_layers, pin_weight, enable_cxl, pack=0):
        """as LiaOPTModel.place: idempotent per flag set, re-places the layers when the flags change"""
        key = (n_gpu_layers, bool(pin_weight), bool(enable_cxl), LayerStore._fmt_of(pack))
        if self.placed_for == key:
            return
        for i, st in enumerate(self.layers):
            if i < n_gpu_layers:
                st.to_device()
        for i, st in enumerate(self.layers):
            if i < n_gpu_layers:
                continue
            if enable_cxl and pin_weight:
                st.to_cxl(pack)
            elif pin_weight:
                st.to_pinned(pack)
            elif st.tier == "device" or st.packed:
                st.to_pageable()
        torch.cuda.synchronize()
        self.placed_for = key

    def close(self):
        for st in self.layers:
            st.close()


class LlamaKVState:
    def __init__(self, model, B, smax):
        sh = model.shape
        self.B, self.smax, self.len = B, smax, 0
        self.serial, self.version = next(_KV_SERIAL), 0
        self.tensors, self.kv = [], []
        for _ in range(sh.layers):
            k = torch.empty((smax, B, sh.kv_heads, sh.head_dim), dtype=torch.bfloat16, device="cuda")
            v = torch.empty_like(k)
            self.tensors.append((k, v))
            self.kv.append(N.KV(k.data_ptr(), v.data_ptr(), smax, B, 1))


class LlamaScheduler:
    """forward(ids, kv, gpu_percentage=..., num_minibatch=...) -> (logits, next ids) on the device."""

    def __init__(self, model, device=0, n_slots=4, pack=None):
        from .scheduler import default_stream_format
        self.model, self.device, self.n_slots = model, device, n_slots
        fmt = default_stream_format() if pack is None else pack
        from .scheduler import wire_format_code
        self.pack = wire_format_code(fmt)
        self.ctx = self.pipe = None
        self.hidden, self.resident, self.tables = {}, {}, None
        self.prefill_tail = True      # last layer of a prefill: last position only behind q|k|v (False: every position)

    def _ensure(self, rows, B, T, n_gpu, smax):
        sh, lib = self.model.shape, N.lib()
        lm_bytes = 2 * 256 * sh.hidden + 8 * min(B, 256) * sh.vocab * 4 + (1 << 20)
        need = max(lib.lia_llama_workspace_bytes(ctypes.byref(self.model.desc), rows), lm_bytes)
        if self.ctx is None or need > self.ctx.workspace_bytes:
            if self.pipe:
                self.pipe.close()
                self.pipe = None
            if self.ctx:
                self.ctx.close()
            self.ctx = ops.Context(self.device, need)
            from . import hostinfo
            hostinfo.cap_torch_threads()      # the token loop's CPU tensor ops: see hostinfo.cap_torch_threads
        if self.pipe is None and n_gpu < sh.layers:
            self.pipe = WeightPipeline(self.ctx, self.model, self.n_slots)
        if self.tables is None or self.tables[0].shape[0] < smax:
            self.tables = rope_tables(max(smax, 64), sh.head_dim, sh.rope_theta)
        if (B, T) not in self.hidden:
            if len(self.hidden) > 4:
                self.hidden.clear()
            self.hidden[(B, T)] = (torch.empty((B, T, sh.hidden), dtype=torch.bfloat16, device="cuda"),
                                   torch.empty((B, T, sh.hidden), dtype=torch.bfloat16, device="cuda"))
        return self.hidden[(B, T)]

    def _ptrs(self, base):
        return (ctypes.c_void_p * 9)(*[base + o for o in self.model.offsets])

    def forward(self, input_ids, kv_state, gpu_percentage=100, num_minibatch=1, pin_weight=True, enable_cxl=False,
                suppress_token=-1, **_ignored):
        m, sh = self.model, self.model.shape
        lib = N.lib()
        B, T = input_ids.shape
        L = sh.layers
        n_gpu = L if gpu_percentage >= 100 else int(L * gpu_percentage / 100)
        from .scheduler import minibatch_bounds
        minis = minibatch_bounds(B, num_minibatch) if T > 1 else [(0, B)]      # ragged last minibatch, num_minibatch clamped to B
        mini = max(n for _, n in minis)
        if m.placed_for != (n_gpu, bool(pin_weight), bool(enable_cxl), LayerStore._fmt_of(self.pack)):
            if self.pipe is not None:
                self.pipe.drain()          # copies in flight read the host buffers a re-placement frees
            self.resident.clear()
            self._run_key = None
        m.place(n_gpu, pin_weight, enable_cxl, self.pack)
        x, y = self._ensure(mini * T, B, T, n_gpu, kv_state.smax)
        ctx, pipe = self.ctx, self.pipe
        st = ctypes.c_void_p(ctx.stream)
        pos0 = kv_state.len
        cos, sin = self.tables
        ids_dev = input_ids.to("cuda").contiguous()
        N.check(lib.lia_llama_embed(ctypes.c_void_p(ids_dev.data_ptr()), ctypes.c_void_p(m.embed_tokens.data_ptr()),
                                    ctypes.c_void_p(x.data_ptr()), B, T, sh.hidden, st), "lia_llama_embed")
        if n_gpu < L:
            pipe.prefetch(n_gpu)
        def resident(i):
            if i not in self.resident:
                self.resident[i] = self._ptrs(m.layers[i].device_ptr())
            return self.resident[i]

        xlast = None
        first = 0
        if T == 1 and n_gpu > 0 and mini == B:
            # decode: the resident run in ONE library call (lia_llama_decode_layers: the per-op route layer by layer with the chained
            # norms, or with LIA_FUSED_DECODE=1 an attention launch and a persistent chain launch per layer, lia_chain.hip)
            key = (n_gpu, kv_state.serial, kv_state.version, resident(0)[0])
            if getattr(self, "_run_key", None) != key:
                ptrs = []
                for i in range(n_gpu):
                    ptrs.extend(resident(i))
                self._run = ((ctypes.c_void_p * (9 * n_gpu))(*ptrs), (ctypes.POINTER(N.KV) * n_gpu)(*[ctypes.pointer(kv_state.kv[i]) for i in range(n_gpu)]))
                self._run_key = key
            N.check(lib.lia_llama_decode_layers(ctx.handle, ctypes.byref(m.desc), n_gpu, ctypes.cast(self._run[0], ctypes.c_void_p),
                                                ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(y.data_ptr()), ctypes.cast(self._run[1], ctypes.c_void_p),
                                                ctypes.c_void_p(cos.data_ptr()), ctypes.c_void_p(sin.data_ptr()), B, pos0, st), "lia_llama_decode_layers")
            x, y = y, x
            first = n_gpu
        for idx in range(first, L):
            if idx < n_gpu:
                w = resident(idx)
                if T == 1 and idx + 1 < n_gpu:
                    ctx.chain_next_norm(resident(idx + 1)[0])      # decode: the next layer's input RMSNorm rides in this layer's down-proj combine
            else:
                w = self._ptrs(pipe.slot_ptrs[self._acquire(pipe, idx)])
                nxt = idx
                for _ in range(pipe.n_slots - 1):
                    nxt = nxt + 1 if nxt + 1 < L else n_gpu
                    if nxt == idx or not pipe.can_prefetch():
                        break
                    pipe.prefetch(nxt)
            tail = self.prefill_tail and T > 1 and pos0 == 0 and idx == L - 1
            if tail:
                xlast = torch.empty((B, 1, sh.hidden), dtype=torch.bfloat16, device="cuda")
            fn = lib.lia_llama_layer_forward_last if tail else lib.lia_llama_layer_forward
            for b0, nb in minis:
                sl = slice(b0, b0 + nb)
                N.check(fn(ctx.handle, ctypes.byref(m.desc), ctypes.byref(w), ctypes.c_void_p(x[sl].data_ptr()),
                           ctypes.c_void_p((xlast if tail else y)[sl].data_ptr()), ctypes.byref(kv_state.kv[idx]),
                           ctypes.c_void_p(cos.data_ptr()), ctypes.c_void_p(sin.data_ptr()), nb, T, pos0, b0, st),
                        "lia_llama_layer_forward")
            if idx >= n_gpu:
                pipe.release(idx)
            x, y = y, x
        logits = torch.empty((B, sh.vocab), dtype=torch.bfloat16, device="cuda")
        nxt_ids = torch.empty((B,), dtype=torch.int64, device="cuda")
        x_lm, T_lm = (x, T) if xlast is None else (xlast, 1)     # (the last layer already reduced the block to its last position)
        N.check(lib.lia_llama_lm_head(ctx.handle, ctypes.c_void_p(x_lm.data_ptr()), B, T_lm, sh.hidden, ctypes.c_void_p(m.final_norm_w.data_ptr()),
                                      ctypes.c_void_p(m.lm_head.data_ptr()), sh.vocab, sh.rms_eps, suppress_token,
                                      ctypes.c_void_p(logits.data_ptr()), ctypes.c_void_p(nxt_ids.data_ptr()), st), "lia_llama_lm_head")
        ctx.synchronize()
        kv_state.len = pos0 + T
        return logits, nxt_ids

    @staticmethod
    def _acquire(pipe, idx):
        pipe.acquire(idx)
        return pipe.held[idx]

    def stream_stats(self, reset=False):
        return self.pipe.stats(reset) if self.pipe else (0.0, 0.0)

    def close(self):
        if self.pipe:
            self.pipe.close()
            self.pipe = None
        if self.ctx:
            self.ctx.close()
            self.ctx = None
