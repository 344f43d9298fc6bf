"""OPT model container for the LIA hot path: shapes, weight tiers (HBM-resident / pinned host /
NUMA-CXL host / pageable host) and the packed per-layer wire format.

Reference counterparts:
  * shapes                      llm/utils/opt-weight-gen.py:84-96 (+ HF OPT configs)
  * move_gpu_layer              lia/modeling_opt.py:229-268   -> LayerStore.to_device()
  * pin_memory (pinned / CXL)   lia/modeling_opt.py:167-227   -> LayerStore.to_pinned() / to_cxl()
  * create_buffer's 16 tensors  lia/modeling_opt.py:90-126    -> one packed flat buffer per layer
  * dummy weights               llm/utils/opt-weight-gen.py:61-69 (torch.rand_like, unseeded) -> init="uniform01", seeded
"""
import ctypes
from dataclasses import dataclass

import numpy as np
import torch

from . import _native as N
from . import ops

LAYER_TENSORS = ops.LAYER_TENSORS


@dataclass(frozen=True)
class OPTShape:
    name: str
    hidden: int
    heads: int
    ffn: int
    layers: int
    vocab: int = 50272
    max_pos: int = 2048
    ln_eps: float = 1e-5

    @property
    def head_dim(self):
        return self.hidden // self.heads

    def layer_param_bytes(self):
        H, F = self.hidden, self.ffn
        return 2 * (4 * H * H + 2 * H * F + 9 * H + F)


# do_layer_norm_before=True family only (opt-350m is post-LN with a 512-wide projection: out of scope)
SHAPES = {s.name: s for s in (
    OPTShape("opt-125m", 768, 12, 3072, 12),
    OPTShape("opt-1.3b", 2048, 32, 8192, 24),
    OPTShape("opt-2.7b", 2560, 32, 10240, 32),
    OPTShape("opt-6.7b", 4096, 32, 16384, 32),
    OPTShape("opt-13b", 5120, 40, 20480, 40),
    OPTShape("opt-30b", 7168, 56, 28672, 48),
    OPTShape("opt-66b", 9216, 72, 36864, 64),
    OPTShape("opt-175b", 12288, 96, 49152, 96),
)}


def resolve_shape(name):
    key = name.lower().split("/")[-1]
    if key not in SHAPES:
        raise ValueError(f"unknown OPT shape {name!r}; known: {sorted(SHAPES)}")
    return SHAPES[key]


def _drawer(init, g):
    def draw(*size):          # ("trained-like": embeddings and every tensor draw_layer does not special-case are N(0, 0.02))
        if init == "uniform01":
            return torch.rand(*size, generator=g, device="cuda", dtype=torch.float32).to(torch.bfloat16)
        return (0.02 * torch.randn(*size, generator=g, device="cuda", dtype=torch.float32)).to(torch.bfloat16)
    return draw


def draw_head(shape, seed=0, init="normal"):
    """embed_tokens, embed_positions, final LN weight / bias of a seeded random model (see LiaOPTModel.random_init)"""
    g = torch.Generator(device="cuda")
    g.manual_seed(seed * 100003)
    draw = _drawer(init, g)
    H = shape.hidden
    tok, pos = draw(shape.vocab, H), draw(shape.max_pos + 2, H)
    if init == "uniform01":      # opt-weight-gen.py:61-62 draws EVERY parameter, LayerNorm weights and biases included
        return tok, pos, draw(H), draw(H)
    return tok, pos, torch.ones(H, dtype=torch.bfloat16, device="cuda"), torch.zeros(H, dtype=torch.bfloat16, device="cuda")


# init="trained-like" (r05, build-defined stress of the wire format; no checkpoint is available offline): what a trained OPT
# layer looks like to an exponent coder and a N(0, 0.02) draw does not -- every tensor has its own scale (spread over ~3 binades,
# fc2 / out-proj smaller than q / k / v, a per-layer drift), 0.1 % of a weight matrix's input channels are outlier columns at 20
# sigma (the "massive activation" channels of trained OPT models), LayerNorm gains sit near 1 with a few large ones, biases and
# LayerNorm offsets are small but not zero.
_TRAINED_SCALE = {"q_w": 1.6, "k_w": 1.6, "v_w": 0.9, "out_w": 0.7, "fc1_w": 1.0, "fc2_w": 0.45}


def _draw_trained_like(n, r, c, g, li, layers):
    sigma = 0.02 * _TRAINED_SCALE[n] * (2.0 ** (-1.5 * li / max(1, layers - 1))) * (0.75 + 0.5 * float(torch.rand(1, generator=g, device="cuda")))
    w = sigma * torch.randn(r, c, generator=g, device="cuda", dtype=torch.float32)
    n_out = max(1, int(round(0.001 * c)))
    cols = torch.randperm(c, generator=g, device="cuda")[:n_out]
    w[:, cols] = 20.0 * sigma * torch.randn(r, n_out, generator=g, device="cuda", dtype=torch.float32).sign() * \
        (0.8 + 0.4 * torch.rand(r, n_out, generator=g, device="cuda"))
    return w.to(torch.bfloat16)


def draw_layer(shape, offsets, layer_bytes, li, seed=0, init="normal"):
    """Layer li of a seeded random model as the packed flat buffer (CUDA bf16): every layer has its own seed, so any subset of
    layers -- one at a time in the checkpoint writer, the resident prefix on a data-parallel peer -- draws the same values."""
    g = torch.Generator(device="cuda")
    g.manual_seed(seed * 100003 + li + 1)
    draw = _drawer(init, g)
    H, F = shape.hidden, shape.ffn
    shapes = {"q_w": (H, H), "k_w": (H, H), "v_w": (H, H), "out_w": (H, H), "fc1_w": (F, H), "fc2_w": (H, F)}
    flat = torch.zeros(layer_bytes // 2, dtype=torch.bfloat16, device="cuda")
    for i, n in enumerate(LAYER_TENSORS):
        o = offsets[i] // 2
        if n in shapes:
            r, c = shapes[n]
            flat[o:o + r * c] = (_draw_trained_like(n, r, c, g, li, shape.layers) if init == "trained-like" else draw(r, c)).reshape(-1)
        elif init == "uniform01":
            k = F if n == "fc1_b" else H
            flat[o:o + k] = draw(k)
        elif init == "trained-like":
            k = F if n == "fc1_b" else H
            if n in ("ln1_w", "ln2_w"):
                gam = 1.0 + 0.15 * torch.randn(k, generator=g, device="cuda")
                big = torch.randperm(k, generator=g, device="cuda")[:max(1, k // 512)]
                gam[big] = 4.0 + 4.0 * torch.rand(big.numel(), generator=g, device="cuda")
                flat[o:o + k] = gam.to(torch.bfloat16)
            else:
                flat[o:o + k] = (0.05 * torch.randn(k, generator=g, device="cuda")).to(torch.bfloat16)
        elif n in ("ln1_w", "ln2_w"):
            flat[o:o + H] = 1.0
    return flat


class LayerStore:
    """One decoder layer's 16 tensors packed into a flat 256-byte-aligned buffer (lia_layer_pack_offsets),
    living in exactly one tier at a time."""

    def __init__(self, desc, offsets, total_bytes):
        self.desc, self.offsets, self.nbytes = desc, offsets, total_bytes
        self.tier = None          # "device" | "pinned" | "cxl" | "mapped" | "pageable" | "remote"
        self.packed = 0           # 0: host copy is raw bf16; 10: it holds the lossless pack10 encoding (lia_pack10.hip)
        self.shard = None         # (rank, world, slice bytes) when the host copy is one slice of the wire bytes
        self.stream_bytes = total_bytes   # bytes that cross the host link per use
        self.want_fmt = 0         # wire format asked for by the last to_pinned / to_cxl (a layer that does not fit stays raw)
        self._dev = None          # torch uint8 CUDA tensor
        self._np = None           # numpy uint8 (pageable)
        self._ptr = None          # raw host pointer (pinned / cxl)
        self._lib = N.lib()

    # -- construction -----------------------------------------------------------------------------
    def set_from_device(self, flat_u8):
        assert flat_u8.is_cuda and flat_u8.dtype == torch.uint8 and flat_u8.numel() == self.nbytes
        self._free()
        self._dev, self.tier = flat_u8, "device"

    def set_from_numpy(self, tensors):
        """tensors: dict name -> uint16 ndarray (bf16 bits), row-major linears."""
        flat = np.zeros(self.nbytes, np.uint8)
        for i, n in enumerate(LAYER_TENSORS):
            a = np.ascontiguousarray(tensors[n], dtype=np.uint16).reshape(-1).view(np.uint8)
            flat[self.offsets[i]: self.offsets[i] + a.size] = a
        self._free()
        self._np, self.tier = flat, "pageable"

    # -- tier moves -------------------------------------------------------------------------------
    # Every move goes through ONE intermediate, the raw bf16 layer in device memory (_raw_on_device): whatever tier and wire
    # format the layer is in, it can be re-placed in any other, so one model object serves calls with different policies,
    # gpu% and wire formats (the reference re-tiers with move_gpu_layer / pin_memory on every first forward of a process,
    # lia/modeling_opt.py:1182-1184,1214-1217; its scripts start one process per flag set).
    def _host_view(self):
        if self.tier == "pageable":
            return self._np
        if self.tier in ("pinned", "cxl", "mapped") and not self.packed and not self.shard:
            return np.ctypeslib.as_array((ctypes.c_uint8 * self.nbytes).from_address(self._ptr))
        raise RuntimeError(f"no raw host view of a layer in tier {self.tier!r} (packed={self.packed}, shard={self.shard})")

    def host_ptr(self):
        if self.tier == "pageable":
            return self._np.ctypes.data
        if self.tier in ("pinned", "cxl", "mapped"):
            return self._ptr
        raise RuntimeError("layer is on the device")

    def raw_host_ptr(self):
        """raw bf16 host copy for the host cores (policy 1 / a host-computed layer): the second, raw pinned copy when the
        streamed copy is packed, else the host copy itself; None when there is none"""
        if getattr(self, "_raw_ptr", None):
            return self._raw_ptr
        return None if (self.packed or self.shard or self.tier not in ("pinned", "cxl", "mapped", "pageable")) else self.host_ptr()

    def device_ptr(self):
        assert self.tier == "device"
        return self._dev.data_ptr()

    def _raw_on_device(self):
        """The raw bf16 layer as a CUDA uint8 tensor, from whatever tier / wire format holds it now (a packed host copy is
        shipped encoded and rebuilt by the same decode kernels the streamer uses)."""
        if self.tier == "device":
            return self._dev
        if self.tier in (None, "remote"):
            raise RuntimeError(f"layer holds no data (tier {self.tier!r})")
        dev = torch.empty(self.nbytes, dtype=torch.uint8, device="cuda")
        raw = getattr(self, "_raw_ptr", None)
        if raw or not (self.packed or self.shard):
            N.check(self._lib.lia_memcpy_h2d(dev.data_ptr(), raw or self.host_ptr(), self.nbytes), "lia_memcpy_h2d")
            return dev
        if self.shard:
            raise ValueError("this rank holds one slice of the layer's wire bytes (allgather streaming); it cannot be re-tiered here")
        enc = torch.empty(self.stream_bytes, dtype=torch.uint8, device="cuda")
        N.check(self._lib.lia_memcpy_h2d(enc.data_ptr(), self._ptr, self.stream_bytes), "lia_memcpy_h2d")
        N.check(self._lib.lia_pack_decode(ctypes.c_void_p(enc.data_ptr()), ctypes.c_void_p(dev.data_ptr()), self.nbytes // 2,
                                          int(self.packed), None), "lia_pack_decode")
        torch.cuda.synchronize()
        return dev

    def _fill_host(self, ptr, src):
        N.check(self._lib.lia_memcpy_d2h(ptr, src.data_ptr(), self.nbytes), "lia_memcpy_d2h")

    def to_device(self):
        """move_gpu_layer (lia/modeling_opt.py:229-268), minus the un-blocking (weights are already row-major)."""
        if self.tier == "device":
            return
        dev = self._raw_on_device()
        self._free()
        self._dev, self.tier = dev, "device"

    def to_pageable(self):
        """plain (unpinned) host memory: what the reference streams from without --pin-weight (lia/modeling_opt.py:1219-1220)"""
        if self.tier == "pageable":
            return
        src = self._raw_on_device()
        host = np.empty(self.nbytes, np.uint8)
        N.check(self._lib.lia_memcpy_d2h(host.ctypes.data, src.data_ptr(), self.nbytes), "lia_memcpy_d2h")
        self._free()
        self._np, self.tier = host, "pageable"

    def _encode_packed(self, fmt, src):
        """src: the raw layer on the device -> (device uint8 tensor with the encoded bytes, n bytes), or None when the layer
        does not fit the format."""
        if fmt in (10, 11):
            if self.nbytes % 2048:
                return None
        elif fmt == 12:
            if self.nbytes % 32:
                return None
        else:
            return None
        if fmt != 10:
            raise ValueError(f"unknown wire format {fmt!r}: 0 (raw bf16) or 10 (pack10)")
        bound, encode = self._lib.lia_pack10_bound, self._lib.lia_pack10_encode
        cap = bound(self.nbytes // 2)
        enc = torch.empty(cap, dtype=torch.uint8, device="cuda")
        out = ctypes.c_size_t()
        rc = encode(ctypes.c_void_p(src.data_ptr()), self.nbytes // 2, ctypes.c_void_p(enc.data_ptr()), cap, ctypes.byref(out))
        if rc == 1:
            return None            # too many out-of-window values: ship this layer raw
        if rc != 0:
            raise N.LiaHipError(f"lia_pack{fmt}_encode failed ({rc})")
        if out.value >= self.nbytes:
            return None            # the encoding is no smaller than the raw layer (very wide distribution): ship raw
        return enc, out.value

    @staticmethod
    def _fmt_of(pack):
        return {False: 0, True: 10, None: 0, "raw": 0, "pack10": 10}.get(pack, pack)          # accepts False / True (= 10) / 0 / 10 / the names

    def to_pinned(self, wire=False, shard=None, keep_raw=False):
        """Tensor.pin_memory() for all 16 tensors at once (lia/modeling_opt.py:207-227); with a packed format the pinned copy
        is that lossless encoding (67-75 % of the bytes).  shard = (r, G): keep only the r-th of G equal slices of the wire
        bytes (data-parallel "allgather" streaming: every rank pulls its slice over its own link).  keep_raw (with a packed
        format): also keep a raw pinned copy for the host cores (a host-computed layer streams packed in the prefill and is
        read raw in decode); dropped silently when the container has no room for it -- the layer then stays raw only.
        A layer already pinned in ANOTHER format (or tier) is re-encoded: nothing is sticky."""
        fmt = self._fmt_of(wire)
        if shard is not None and shard[1] > 1:
            if self.tier == "pinned" and self.shard and self.shard[:2] == tuple(shard) and self.want_fmt == fmt:
                return
            return self._to_pinned_shard(fmt, shard)
        if (self.tier == "pinned" and not self.shard and self.want_fmt == fmt and
                (not keep_raw or not self.packed or getattr(self, "_raw_ptr", None))):
            return
        src = self._raw_on_device()
        self._free()                                   # the host copy goes first: never two generations of a layer in host memory
        from . import hostinfo
        raw_ptr = ptr = None
        try:
            enc = self._encode_packed(fmt, src)
            if enc and keep_raw:
                try:
                    hostinfo.guard_host_allocation(self.nbytes + enc[1], "raw + packed pinned copies of a host-computed layer", ceiling=0.85)
                    raw_ptr = self._lib.lia_host_alloc_pinned(self.nbytes)
                except MemoryError:
                    raw_ptr = None
                if raw_ptr:
                    self._fill_host(raw_ptr, src)
                else:
                    enc = None               # no room for two copies: keep the raw one only (it also streams, just more bytes)
            nbytes = enc[1] if enc else self.nbytes
            hostinfo.guard_host_allocation(nbytes, "pinning a streamed layer")
            ptr = self._lib.lia_host_alloc_pinned(nbytes)
            if not ptr:
                raise MemoryError("Fail to allocate pinned memory: " + self._lib.lia_last_error().decode())
            if enc:
                N.check(self._lib.lia_memcpy_d2h(ptr, enc[0].data_ptr(), nbytes), "lia_memcpy_d2h")
            else:
                self._fill_host(ptr, src)
        except BaseException:
            # nothing is lost: whatever failed (the guard, the allocation, the copy), the layer stays where it can be rebuilt from
            for p_ in (raw_ptr, ptr):
                if p_:
                    self._lib.lia_host_free_pinned(p_)
            self._dev, self.tier = src, "device"
            raise
        self._ptr, self.tier, self._raw_ptr = ptr, "pinned", raw_ptr
        self.packed, self.stream_bytes, self.want_fmt = (fmt if enc else 0), nbytes, fmt

    @staticmethod
    def shard_bytes(total, world):
        """slice size of a `total`-byte wire buffer over `world` ranks: equal, 256-byte aligned slices (the last one padded)"""
        return ((total + world - 1) // world + 255) // 256 * 256

    def _to_pinned_shard(self, fmt, shard):
        r, G = shard
        src0 = self._raw_on_device()
        self._free()
        ptr = None
        try:
            enc = self._encode_packed(fmt, src0)
            if enc:
                src, total = enc
            else:
                src, total = src0, self.nbytes
            sh = self.shard_bytes(total, G)
            lo, hi = min(r * sh, total), min((r + 1) * sh, total)
            from . import hostinfo
            hostinfo.guard_host_allocation(sh, "pinning a slice of a streamed layer")
            ptr = self._lib.lia_host_alloc_pinned(sh)
            if not ptr:
                raise MemoryError("Fail to allocate pinned memory: " + self._lib.lia_last_error().decode())
            ctypes.memset(ptr, 0, sh)
            if hi > lo:
                N.check(self._lib.lia_memcpy_d2h(ptr, src.data_ptr() + lo, hi - lo), "lia_memcpy_d2h")
        except BaseException:
            if ptr:
                self._lib.lia_host_free_pinned(ptr)
            self._dev, self.tier = src0, "device"      # the whole raw layer is still here: nothing is lost
            raise
        self._ptr, self.tier = ptr, "pinned"
        self.packed, self.stream_bytes, self.shard, self.want_fmt = (fmt if enc else 0), total, (r, G, sh), fmt

    def to_cxl(self, pack=0):
        """realloc_to_numa (lia/modeling_opt.py:168-175) + hipHostRegister so the copy engine can DMA from it
        (the reference leaves the CXL copy pageable, lia/cxl/numa_alloc.py:49).  pack = 10: the tier holds
        that lossless wire format instead of raw bf16 (fewer bytes in the tier AND on the link)."""
        fmt = self._fmt_of(pack)
        if self.tier == "cxl" and self.want_fmt == fmt:
            return
        src = self._raw_on_device()
        self._free()
        ptr, registered, nbytes = None, False, self.nbytes
        try:
            enc = self._encode_packed(fmt, src)
            nbytes = enc[1] if enc else self.nbytes
            from . import hostinfo
            hostinfo.guard_host_allocation(nbytes, "CXL-tier copy of a streamed layer")
            ptr = self._lib.numa_alloc_interleave(nbytes)
            if not ptr:
                raise MemoryError("Fail to allocate CXL memory!")  # same text as lia/modeling_opt.py:175
            N.check(self._lib.lia_numa_register(ptr, nbytes), "lia_numa_register")   # register first: the fill below then runs at DMA speed
            registered = True
            if enc:
                N.check(self._lib.lia_memcpy_d2h(ptr, enc[0].data_ptr(), nbytes), "lia_memcpy_d2h")
            else:
                self._fill_host(ptr, src)
        except BaseException:
            if registered:
                self._lib.lia_numa_unregister(ptr)
            if ptr:
                self._lib.numa_free_node(ptr, nbytes)
            self._dev, self.tier = src, "device"
            raise
        self._ptr, self.tier = ptr, "cxl"
        self.packed, self.stream_bytes, self.want_fmt = (fmt if enc else 0), nbytes, fmt

    def set_from_mapped_file(self, path, offset, nbytes, fmt):
        """The layer's wire bytes straight from a checkpoint file of the build's on-disk format (lia_amd.packed_checkpoint):
        an mmap of the file range registered with the driver (hipHostRegister) so the copy engine DMAs from the page cache.
        First choice is a READ-ONLY SHARED mapping registered read-only (lia_numa_register_readonly): pinning without write
        intent leaves the page-cache pages where they are -- no second host copy.  A driver that refuses that gets a private
        writable mapping (r02's form; pinning it with write intent may copy the pages into anonymous memory, so the guard
        below counts the layer as a full allocation either way).  fmt = 0 (raw bf16) or the packed format the file holds."""
        import mmap
        if fmt not in (0, 10):
            raise ValueError(f"{path}: wire format {fmt} is not supported by this build (raw or pack10)")
        if fmt == 0 and nbytes != self.nbytes:
            raise ValueError(f"{path}: raw layer of {nbytes} bytes, expected {self.nbytes}")
        from . import hostinfo
        hostinfo.guard_host_allocation(nbytes, f"mapping + registering {path}")
        self._free()
        gran = mmap.ALLOCATIONGRANULARITY
        start = offset // gran * gran
        length = nbytes + (offset - start)
        mm = base = mode = None
        with open(path, "rb") as f:
            for mode, access, register in (("shared-readonly", mmap.ACCESS_READ, self._lib.lia_numa_register_readonly),
                                           ("private", mmap.ACCESS_COPY, self._lib.lia_numa_register)):
                mm = mmap.mmap(f.fileno(), length, access=access, offset=start)
                base = np.frombuffer(mm, dtype=np.uint8).ctypes.data          # (works for read-only buffers, unlike ctypes.from_buffer)
                rc = register(base, length)
                if rc == 0:
                    break
                try:
                    mm.close()
                except BufferError:
                    pass
                mm = None
        if mm is None:
            N.check(rc, f"hipHostRegister of {path}")
        self._mm, self._mm_base, self.map_mode = mm, base, mode
        self._ptr, self.tier = base + (offset - start), "mapped"
        self.packed, self.stream_bytes, self.want_fmt = int(fmt), nbytes, int(fmt)
        if fmt:
            try:
                self._validate_packed(self._ptr, nbytes, path)       # the decode kernel trusts the header: check it before the first stream
            except ValueError:
                self._free()
                raise

    def _validate_packed(self, host_ptr, nbytes, what):
        rc = self._lib.lia_pack10_validate(ctypes.c_void_p(host_ptr), nbytes, self.nbytes // 2)
        if rc != 0:
            why = {-1: "shorter than a header", -2: "bad magic / version", -3: "value count differs from this model's layer",
                   -4: "an offset points outside the buffer", -5: "a count exceeds its capacity"}.get(rc, f"code {rc}")
            raise ValueError(f"{what}: not a consistent pack10 layer of {self.nbytes // 2} values ({why}); the file is stale or corrupt")

    def set_from_file_to_device(self, path, offset, nbytes, fmt):
        """A RESIDENT layer of a packed checkpoint directory: plain read of the wire bytes, one H2D copy, decoded on the device when
        the file holds a packed format.  (r02 mapped + registered + freed every resident layer on its way to HBM.)"""
        if fmt == 0 and nbytes != self.nbytes:
            raise ValueError(f"{path}: raw layer of {nbytes} bytes, expected {self.nbytes}")
        host = np.fromfile(path, dtype=np.uint8, count=nbytes, offset=offset)
        if host.size != nbytes:
            raise ValueError(f"{path}: short read ({host.size} of {nbytes} bytes)")
        if fmt not in (0, 10):
            raise ValueError(f"{path}: wire format {fmt} is not supported by this build (raw or pack10)")
        if fmt:
            self._validate_packed(host.ctypes.data, nbytes, path)
        dev = torch.empty(self.nbytes, dtype=torch.uint8, device="cuda")
        if fmt:
            enc = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
            N.check(self._lib.lia_memcpy_h2d(enc.data_ptr(), host.ctypes.data, nbytes), "lia_memcpy_h2d")
            N.check(self._lib.lia_pack_decode(ctypes.c_void_p(enc.data_ptr()), ctypes.c_void_p(dev.data_ptr()), self.nbytes // 2, int(fmt),
                                              None), "lia_pack_decode")
            torch.cuda.synchronize()
        else:
            N.check(self._lib.lia_memcpy_h2d(dev.data_ptr(), host.ctypes.data, nbytes), "lia_memcpy_h2d")
        self.set_from_device(dev)

    def is_dma_able(self):
        return self.tier in ("pinned", "cxl", "mapped")

    def _free(self):
        if getattr(self, "_raw_ptr", None):
            self._lib.lia_host_free_pinned(self._raw_ptr)
        self._raw_ptr = None
        if self.tier == "pinned" and self._ptr:
            self._lib.lia_host_free_pinned(self._ptr)
        elif self.tier == "cxl" and self._ptr:
            self._lib.lia_numa_unregister(self._ptr)
            self._lib.numa_free_node(self._ptr, self.stream_bytes)      # the size it was allocated with
        elif self.tier == "mapped" and getattr(self, "_mm", None) is not None:
            self._lib.lia_numa_unregister(self._mm_base)
            try:
                self._mm.close()
            except BufferError:
                pass                      # a ctypes view still points into it; the mapping goes with the last reference
            self._mm = None
        self._ptr = self._dev = self._np = None
        self.tier = None
        self.packed, self.stream_bytes, self.shard, self.want_fmt = 0, self.nbytes, None, 0

    def close(self):
        self._free()

    def __del__(self):
        try:
            self._free()
        except Exception:
            pass


class LiaOPTModel:
    """Weights of an OPT decoder-only LM laid out for the offload scheduler.

    embed_tokens (tied lm_head), embed_positions and the final LayerNorm always live in HBM (the
    reference keeps them on the CPU, lia/modeling_opt.py:1108,1563 + models.py:430)."""

    def __init__(self, shape):
        self.shape = shape
        self.desc = ops.make_desc(shape.hidden, shape.heads, shape.ffn, shape.ln_eps)
        self.offsets, self.layer_bytes = ops.pack_offsets(self.desc)
        self.layers = [LayerStore(self.desc, self.offsets, self.layer_bytes) for _ in range(shape.layers)]
        self.embed_tokens = self.embed_positions = self.final_ln_w = self.final_ln_b = None
        self.placed_for = None  # _place_key(...) of the last placement

    # -- builders ---------------------------------------------------------------------------------
    @classmethod
    def from_numpy(cls, shape, m):
        """m: dict from tests/golden/synth.make_model (uint16 bf16 bit arrays)."""
        self = cls(shape)

        def dev(a):
            return torch.from_numpy(np.ascontiguousarray(a).view(np.int16)).view(torch.bfloat16).cuda()

        self.embed_tokens, self.embed_positions = dev(m["embed_tokens"]), dev(m["embed_positions"])
        self.final_ln_w, self.final_ln_b = dev(m["final_ln_w"]), dev(m["final_ln_b"])
        for st, lw in zip(self.layers, m["layers"]):
            st.set_from_numpy(lw)
        return self

    @classmethod
    def random_init(cls, shape, seed=0, init="normal", n_gpu_layers=0, pin_weight=True, enable_cxl=False,
                    host_owner=True, wire=False, raw_layers=(), shard=None):
        """Random-init weights of the exact architecture, generated ON THE GPU one layer at a time and
        moved straight to their tier (an OPT-30B would take minutes to draw on the CPU).
        init="normal": HF _init_weights (lia/modeling_opt.py:895-904): Linear/Embedding ~ N(0, 0.02), zero
        bias, LN = (1, 0).  init="uniform01": the reference's dummy recipe (opt-weight-gen.py:61-62: torch.rand_like on EVERY
        parameter -- weights, biases, LayerNorm weights and biases, embeddings), seeded.
        Every tensor group has its own seed, so data-parallel ranks draw identical resident layers;
        host_owner=False (non-root DP ranks) skips the streamed layers, which arrive by broadcast.
        raw_layers: layers whose host copy stays raw bf16 whatever the wire format (the host cores compute them)."""
        self = cls(shape)
        if host_owner:
            from . import hostinfo
            hostinfo.check_host_allocation(int(self.streamed_bytes(n_gpu_layers) * (0.69 if wire else 1.0) / (shard[1] if shard else 1)),
                                           f"{shape.name}: {shape.layers - n_gpu_layers} streamed layers")
        self.embed_tokens, self.embed_positions, self.final_ln_w, self.final_ln_b = draw_head(shape, seed, init)
        for li, st in enumerate(self.layers):
            if li >= n_gpu_layers and not host_owner:
                st.tier = "remote"          # lives on the DP root's host; reaches this rank by broadcast
                continue
            flat = draw_layer(shape, self.offsets, self.layer_bytes, li, seed, init)
            st.set_from_device(flat.view(torch.uint8))
            if li >= n_gpu_layers:
                fmt = LayerStore._fmt_of(wire)
                if enable_cxl and pin_weight:        # the reference consults enable_cxl only inside pin_memory (:1214-1217)
                    st.to_cxl(0 if li in raw_layers else fmt)
                elif pin_weight:
                    st.to_pinned(fmt, shard=shard, keep_raw=(li in raw_layers))
                else:
                    st.to_pageable()
        torch.cuda.synchronize()
        self.placed_for = self._place_key(n_gpu_layers, pin_weight, enable_cxl, wire, raw_layers, shard)
        return self

    # -- placement (first forward, and again whenever the flags change) -------------------------------
    @staticmethod
    def _place_key(n_gpu_layers, pin_weight, enable_cxl, wire, raw_layers, shard):
        return (int(n_gpu_layers), bool(pin_weight), bool(enable_cxl), LayerStore._fmt_of(wire), frozenset(raw_layers or ()),
                tuple(shard) if shard else None)

    def place(self, n_gpu_layers, pin_weight, enable_cxl, wire=False, raw_layers=(), shard=None):
        """Tier assignment, idempotent per flag set: move_gpu_layer / pin_memory of the reference (lia/modeling_opt.py:1182-1184,
        1214-1217) run on the first forward of a process; here a later call with other flags (policy 1 needs raw host copies,
        another gpu%, another wire format, the CXL tier) RE-PLACES the layers instead of failing -- every LayerStore can be
        rebuilt on the device from whatever it holds."""
        key = self._place_key(n_gpu_layers, pin_weight, enable_cxl, wire, raw_layers, shard)
        if self.placed_for == key:
            return
        fmt = LayerStore._fmt_of(wire)
        from . import hostinfo
        if pin_weight:
            moving = sum(st.nbytes for i, st in enumerate(self.layers) if i >= n_gpu_layers and st.tier in ("pageable", "device"))
            leaving = sum(st.nbytes for i, st in enumerate(self.layers) if i < n_gpu_layers and st.tier == "pageable")
            hostinfo.check_host_allocation(max(0, moving - leaving), "pinning the streamed layers")
        # promotions first (they free host memory), then the host-side moves
        for i, st in enumerate(self.layers):
            if st.tier != "remote" and i < n_gpu_layers:
                st.to_device()
        for i, st in enumerate(self.layers):
            if st.tier == "remote" or i < n_gpu_layers:
                continue
            if enable_cxl and pin_weight:
                st.to_cxl(0 if i in raw_layers else fmt)
            elif pin_weight:
                if st.tier == "mapped" and st.want_fmt == fmt and not shard and i not in raw_layers:
                    continue                  # a registered mapping of the checkpoint file in the wire format asked for IS pinned memory
                st.to_pinned(fmt, shard=shard, keep_raw=(i in raw_layers))
            elif st.tier == "device" or st.packed or st.shard:
                st.to_pageable()              # no --pin-weight: plain host memory, staged through the bounce buffer
            # (a raw pinned / CXL copy left by an earlier --pin-weight call also serves an unpinned request)
        torch.cuda.synchronize()
        self.placed_for = key

    def streamed_bytes(self, n_gpu_layers):
        return (self.shape.layers - n_gpu_layers) * self.layer_bytes

    def close(self):
        for st in self.layers:
            st.close()
