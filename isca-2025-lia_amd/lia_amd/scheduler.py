"""The offload scheduler: MI355X-native counterpart of OPTDecoder.forward (lia/modeling_opt.py:1021-1586).

Same policy table (lia/modeling_opt.py:1167-1176), same flag surface, different machinery:

  reference                                                   here
  ---------------------------------------------------------   ------------------------------------------------------
  16 copy_ per layer on load_weight_stream (:270-293)          ONE pinned hipMemcpyAsync of the packed layer on the
                                                               streamer's copy stream into an HBM slot
  torch.cuda.synchronize() after every minibatch/layer          hipEvent handshakes slot<->compute stream; the host only
  (:1298,1339,1346,1506,1528)                                   blocks for the policy-2 host attention round trip
  first streamed layer loaded synchronously at the start        prefetch runs ahead across token steps (after the last
  of every forward (:1287-1298, :1497-1506)                     layer it wraps to the first streamed layer), so the copy
                                                               engine never idles between forwards
  hidden states bounced through pinned host memory             the whole-batch hidden state stays in HBM (235 MB at
  (:1262-1267, load_activation/store_hidden :320-355)          B=64,T=256,H=7168; the GPU has 288 GB)
  GPU-side un-blocking of every streamed weight (G2)           none: host weights are row-major
  KV cache host tensors re-pinned per call (:1381-1387)        pinned once per generation
  embed / final LN / lm_head on the CPU                        on the GPU (weights resident), last position only

Weights of layers [0, n_gpu) are HBM-resident (policy 3, KV in HBM); layers [n_gpu, L) are streamed and
run the phase's policy: prefill 0 (minibatched, K/V delivered to the host cache) | decode 2 (GPU linears
+ host attention) | decode 0 (cached rows fetched to the GPU).  Policy 1 (all-CPU) is the IPEX baseline
and is not scheduled here.
"""
import ctypes
import itertools
import os

import torch

from . import _native as N
from . import ops


class WeightPipeline:
    """Slot bookkeeping over lia_stream_*: prefetch(layer) -> acquire(layer) -> release(layer)."""

    def __init__(self, ctx, model, n_slots, dp_group=None):
        self.ctx, self.model, self.lib = ctx, model, N.lib()
        self.dp = dp_group
        h = ctypes.c_void_p()
        N.check(self.lib.lia_stream_create(ctx.handle, n_slots, model.layer_bytes, ctypes.byref(h)), "lia_stream_create")
        self.handle, self.n_slots = h, n_slots
        self.free_slots = list(range(n_slots))
        self.inflight = []          # [(layer_idx, slot)] in issue order, not yet acquired
        self.held = {}              # layer_idx -> slot (acquired, not yet released)
        self.slot_ptrs = [self.lib.lia_stream_slot_ptr(h, s) for s in range(n_slots)]
        self.ptr_arrays = [ops.weight_ptr_array(p, model.offsets) for p in self.slot_ptrs]
        if dp_group is not None:
            from .dp import RawDeviceBuffer
            self.slot_tensors = [RawDeviceBuffer(p, model.layer_bytes).tensor() for p in self.slot_ptrs]
            self.copy_stream = torch.cuda.ExternalStream(self.lib.lia_stream_copy_stream(h))
            self.staging_tensors = None
            # the root tells every rank which layers travel pack10-encoded and how many bytes each one ships
            # (gloo validation runs without a GPU -- tests/test_dp_gloo.py -- keep the table on the host)
            meta = torch.zeros((len(model.layers), 2), dtype=torch.int64, device="cuda" if torch.cuda.is_available() else "cpu")
            if dp_group.is_root:
                for i, st in enumerate(model.layers):
                    if st.tier not in ("device", None):
                        meta[i, 0], meta[i, 1] = int(st.packed), int(st.stream_bytes)
            dp_group.dist.broadcast(meta, src=dp_group.root)
            self.layer_meta = meta.cpu().tolist()
            if dp_group.mode == "allgather" and dp_group.world > 1:
                for i, st in enumerate(model.layers):
                    if st.tier not in ("device", None, "remote") and [int(st.packed), int(st.stream_bytes)] != self.layer_meta[i]:
                        raise RuntimeError(f"layer {i}: this rank's wire encoding differs from the root's (allgather streaming needs "
                                           "identical weights and format on every rank)")

    def can_prefetch(self):
        return bool(self.free_slots)

    def prefetch(self, layer_idx):
        if any(li == layer_idx for li, _ in self.inflight) or layer_idx in self.held:
            return
        if not self.can_prefetch():
            return
        st = self.model.layers[layer_idx]
        slot = self.free_slots.pop(0)
        if self.dp is None and st.packed:
            N.check(self.lib.lia_stream_prefetch_packed(self.handle, slot, ctypes.c_void_p(st.host_ptr()), st.stream_bytes, st.nbytes // 2,
                                                        int(st.packed), int(st.is_dma_able())), "lia_stream_prefetch_packed")
        elif self.dp is None:
            N.check(self.lib.lia_stream_prefetch(self.handle, slot, ctypes.c_void_p(st.host_ptr()), st.nbytes, int(st.is_dma_able())),
                    "lia_stream_prefetch")
        elif self.dp.mode == "allgather" and self.dp.world > 1:        # (one rank: nothing to gather, the broadcast path serves)
            self._prefetch_allgather(st, slot, layer_idx)
        else:
            self._prefetch_broadcast(st, slot, layer_idx)
        self.inflight.append((layer_idx, slot))

    def _prefetch_broadcast(self, st, slot, layer_idx):
        """Root: host -> slot in chunks, each chunk RCCL-broadcast over xGMI as soon as it is enqueued, so chunk
        k travels to the peers while chunk k+1 is still arriving over PCIe.  Peers: receive into the same slot.
        Everything is ordered on the copy stream; the slot is declared ready after the last broadcast."""
        from .dp import RawDeviceBuffer, broadcast_chunked
        dp = self.dp
        packed, nbytes = self.layer_meta[layer_idx]
        N.check(self.lib.lia_stream_begin(self.handle, slot), "lia_stream_begin")
        if packed:
            target, copy_fn = self._staging()[slot][:nbytes], self.lib.lia_stream_copy_chunk_packed
        else:
            target, copy_fn = self.slot_tensors[slot][:self.model.layer_bytes], self.lib.lia_stream_copy_chunk
        before = None
        if dp.is_root:
            base, pinned = st.host_ptr(), int(st.is_dma_able())

            def before(off, n):
                N.check(copy_fn(self.handle, slot, off, ctypes.c_void_p(base + off), n, pinned), "lia_stream_copy_chunk")
        with torch.cuda.stream(self.copy_stream):
            works = broadcast_chunked(dp.dist, target, dp.root, dp.chunk_bytes, before_chunk=before)
            for w in works:
                w.wait()              # the copy stream waits for RCCL's stream; the host does not block
        if packed:
            N.check(self.lib.lia_stream_decode_packed(self.handle, slot, self.model.layer_bytes // 2, int(packed)), "lia_stream_decode_packed")
        N.check(self.lib.lia_stream_mark_ready(self.handle, slot), "lia_stream_mark_ready")

    def _staging(self):
        if self.staging_tensors is None:
            from .dp import RawDeviceBuffer
            cap = self.lib.lia_pack10_bound(self.model.layer_bytes // 2)
            self.staging_tensors = [RawDeviceBuffer(self.lib.lia_stream_staging_ptr(self.handle, s), cap).tensor()
                                    for s in range(self.n_slots)]
        return self.staging_tensors

    def _prefetch_allgather(self, st, slot, layer_idx):
        """Every rank copies ITS slice of the layer's wire bytes host -> device over its own link, then one all-gather on
        the copy stream assembles the layer in every rank's slot (or staging area, then the decode kernel)."""
        from .model import LayerStore
        dp = self.dp
        packed, nbytes = self.layer_meta[layer_idx]
        G, r = dp.world, dp.rank
        sh = LayerStore.shard_bytes(nbytes, G)
        if st.shard is None or st.shard != (r, G, sh):
            raise ValueError(f"layer {layer_idx}: allgather streaming needs the host copy pinned as slice {r} of {G} "
                             "(LiaOPTModel.random_init(..., shard=(rank, world)))")
        N.check(self.lib.lia_stream_begin(self.handle, slot), "lia_stream_begin")
        if packed:
            full, copy_fn = self._staging()[slot], self.lib.lia_stream_copy_chunk_packed
        else:
            full, copy_fn = self.slot_tensors[slot], self.lib.lia_stream_copy_chunk
        if G * sh > full.numel():
            raise ValueError(f"layer {layer_idx}: {G} slices of {sh} bytes do not fit the {full.numel()}-byte slot")
        full = full[:G * sh]
        N.check(copy_fn(self.handle, slot, r * sh, ctypes.c_void_p(st.host_ptr()), sh, int(st.is_dma_able())), "lia_stream_copy_chunk")
        mine = full[r * sh:(r + 1) * sh]
        with torch.cuda.stream(self.copy_stream):
            if dp.dist.get_backend() == "nccl":
                dp.dist.all_gather_into_tensor(full, mine, async_op=True).wait()      # in place: `mine` is slice r of `full`
            else:
                parts = [torch.empty_like(mine) for _ in range(G)]                    # gloo (validation runs): out of place
                dp.dist.all_gather(parts, mine.clone())
                for g, p in enumerate(parts):
                    if g != r:
                        full[g * sh:(g + 1) * sh].copy_(p)
        if packed:
            N.check(self.lib.lia_stream_decode_packed(self.handle, slot, self.model.layer_bytes // 2, int(packed)), "lia_stream_decode_packed")
        N.check(self.lib.lia_stream_mark_ready(self.handle, slot), "lia_stream_mark_ready")

    def acquire(self, layer_idx):
        """Make the compute stream wait for the layer's copy; returns the 16 device pointers."""
        pos = next((j for j, (li, _) in enumerate(self.inflight) if li == layer_idx), None)
        if pos is None:
            # the layer was not prefetched (a prefill after a decode step with host-computed layers, a different gpu%):
            # load it on demand.  Queued copies of OTHER layers are kept -- the copy stream runs them in order and they
            # are acquired later in this forward; only when every slot is taken is the NEWEST of them given up
            # (lia_stream_begin orders the slot's next copy behind the dropped one's decode kernel).
            while not self.free_slots and self.inflight:
                _, s_drop = self.inflight.pop()
                self.free_slots.append(s_drop)
            if not self.free_slots:
                raise RuntimeError(f"no streamer slot free for layer {layer_idx}: {len(self.held)} layers are held")
            self.prefetch(layer_idx)
            pos = len(self.inflight) - 1
        li, slot = self.inflight.pop(pos)
        N.check(self.lib.lia_stream_wait(self.handle, slot, ctypes.c_void_p(self.ctx.stream)), "lia_stream_wait")
        self.held[layer_idx] = slot
        return self.ptr_arrays[slot]

    def release(self, layer_idx):
        slot = self.held.pop(layer_idx)
        N.check(self.lib.lia_stream_release(self.handle, slot, ctypes.c_void_p(self.ctx.stream)), "lia_stream_release")
        self.free_slots.append(slot)

    def forget(self, layer_idx):
        """Give up a queued copy of layer_idx (it will not be used: the layer's decode step moved to the host cores).  The slot is
        free again at once; lia_stream_begin orders its next copy behind the abandoned one."""
        for j, (li, slot) in enumerate(self.inflight):
            if li == layer_idx:
                self.inflight.pop(j)
                self.free_slots.append(slot)
                return True
        return False

    def decode_stats(self, reset=False, block=True):
        """wire-format decode kernel since the last reset: {"launches", "ms", "bytes_in", "bytes_out"} (lia_stream_decode_stats);
        block=False: never waits for decodes in flight (lia_stream_poll_decode_stats) -- for the edge of a timed region"""
        n, ms, bi, bo = ctypes.c_long(), ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        fn = self.lib.lia_stream_decode_stats if block else self.lib.lia_stream_poll_decode_stats
        N.check(fn(self.handle, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(bi), ctypes.byref(bo), int(reset)))
        return {"launches": n.value, "ms": ms.value, "bytes_in": bi.value, "bytes_out": bo.value}

    def poll_stats(self):
        """(bytes, busy ms) of the copies that have COMPLETED so far -- never blocks (stats() waits for the queued copies)"""
        b, ms = ctypes.c_double(), ctypes.c_double()
        N.check(self.lib.lia_stream_poll_stats(self.handle, ctypes.byref(b), ctypes.byref(ms)))
        return b.value, ms.value

    def drain(self):
        """Forget every queued copy and wait for the copy engine: called before the model re-tiers its layers."""
        for _, slot in self.inflight:
            self.free_slots.append(slot)
        self.inflight = []
        torch.cuda.synchronize()

    def stats(self, reset=False):
        b, ms = ctypes.c_double(), ctypes.c_double()
        N.check(self.lib.lia_stream_stats(self.handle, ctypes.byref(b), ctypes.byref(ms), int(reset)))
        return b.value, ms.value

    def close(self):
        if self.handle:
            self.lib.lia_stream_destroy(self.handle)
            self.handle = None


class PinnedPool:
    """Exact-size pinned host blocks with a per-size free list.  torch's pin_memory=True rounds every block up to a
    power of two (a 264 MB OPT-30B cache tensor takes 512 MB, 23 GB of host KV become 45 GB) -- on a box whose container
    limit decides which model fits, the caches are allocated here instead and recycled across generate() calls."""
    _free = {}

    @classmethod
    def acquire(cls, nbytes):
        lst = cls._free.get(nbytes)
        if lst:
            return lst.pop()
        from . import hostinfo
        hostinfo.guard_host_allocation(nbytes, "pinned host KV cache")
        ptr = N.lib().lia_host_alloc_pinned(nbytes)
        if not ptr:
            raise MemoryError("Fail to allocate pinned memory: " + N.lib().lia_last_error().decode())
        return ptr

    @classmethod
    def release(cls, ptr, nbytes):
        cls._free.setdefault(nbytes, []).append(ptr)

    @classmethod
    def idle_blocks(cls, nbytes):
        return len(cls._free.get(nbytes, ()))

    @classmethod
    def trim(cls):
        for lst in cls._free.values():
            for ptr in lst:
                N.lib().lia_host_free_pinned(ptr)
        cls._free.clear()

    @staticmethod
    def as_tensor(ptr, shape):
        n = 1
        for d in shape:
            n *= d
        buf = (ctypes.c_char * (2 * n)).from_address(ptr)
        return torch.frombuffer(buf, dtype=torch.int16).view(torch.bfloat16).view(*shape)


COOP_HEADROOM = 10                    # raw host copies kept beyond the planner's count: the room the online controller may search upward
DEFAULT_STREAM_FORMAT = "pack10"      # one default for OffloadScheduler, run_generation.py and bench.py


def default_stream_format():
    """Wire format of the pinned streamed layers when the caller names none: $LIA_STREAM_FORMAT, else pack10 -- a LOSSLESS encoding
    (the device decodes it back to the same bf16 bits, tests/test_gpu_ops.py), so the results are those of the reference's raw
    transfer; a layer whose values do not pack (encoding >= raw size, or too many out-of-window values) is pinned raw by itself
    (LayerStore._encode_packed).  `raw` = what the reference ships (Tensor.pin_memory of the bf16 tensors, modeling_opt.py:207-227)."""
    fmt = os.environ.get("LIA_STREAM_FORMAT", DEFAULT_STREAM_FORMAT).lower()
    if fmt not in ("raw", "pack10"):
        raise ValueError(f"LIA_STREAM_FORMAT={fmt!r}: expected raw or pack10")
    return fmt


def minibatch_bounds(B, num_minibatch):
    """[(first row, rows)] of the prefill minibatches: num_minibatch clamped to [1, B]; every minibatch has int(B / num_minibatch)
    rows (lia/modeling_opt.py:1178) except the last, which also takes the remainder the reference drops."""
    B, n = int(B), max(1, min(int(num_minibatch), int(B)))
    if B <= 0:
        return []
    mini = B // n
    return [(i * mini, mini) for i in range(n - 1)] + [((n - 1) * mini, B - (n - 1) * mini)]


def placement_formats(prefill_policy, decoding_policy, wire, n_gpu, L, pin_weight, enable_cxl, cpu_set=frozenset(), data_parallel=False):
    """(wire format of the streamed layers' host copy, layers that ALSO keep a raw bf16 host copy) for a flag set.  The host cores
    read raw bf16 in place (policy 1, the cooperative split's host layers); the link ships the lossless pack10 bytes.
      * neither phase on the host cores: packed, + raw copies for the cooperative split's candidates (cpu_set);
      * prefill 0 / decode 1 with pinned weights (the README example, llm/scripts/lia_offline.sh:13-19): r06 keeps BOTH -- the
        policy-0 prefill streams every layer once and is link-bound (54.3 GB raw: 950 ms; 36.6 GB packed: 650 ms), the decode
        steps read the raw copies.  +0.675 x the streamed bytes of pinned memory; a layer the container has no room to keep twice
        stays raw only (LayerStore.to_pinned);
      * prefill 1, the NUMA tier (one copy by definition), unpinned weights: raw."""
    if prefill_policy != 1 and decoding_policy != 1:
        return wire, frozenset(cpu_set)
    if wire and prefill_policy in (0, 3) and decoding_policy == 1 and pin_weight and not enable_cxl and not data_parallel:
        return wire, frozenset(range(n_gpu, L))
    return 0, frozenset(cpu_set)


def wire_format_code(fmt):
    """"raw" / "pack10" / 0 / 10 / False / True -> 0 | 10; anything else (pack11 / pack12 of builds before r05) is refused by name"""
    codes = {"raw": 0, "pack10": 10, False: 0, True: 10, 0: 0, 10: 10}
    if fmt not in codes:
        raise ValueError(f"wire format {fmt!r} is not supported by this build (raw or pack10): a model directory written with "
                         "--wire pack11 / pack12 by an earlier build must be rewritten (python -m lia_amd.packed_checkpoint ... --wire pack10)")
    return codes[fmt]


_KV_SERIAL = itertools.count(1)      # KVState / LlamaKVState objects, numbered: id() values come back after a free


class KVState:
    """Per-layer KV caches of one generation: seq-major [Smax,B,h,d] (attentions.py:462-476), in HBM for
    resident layers and in pinned host memory for streamed ones (lia/modeling_opt.py:1270-1281)."""

    def __init__(self, model, n_gpu, B, smax, all_on_device=False, host_layers=(), dual_layers=()):
        """host_layers (with all_on_device): layers whose cache stays in pinned host memory all the same -- the layers the
        cooperative split computes on the host cores (scheduler.forward cpu_layers).  dual_layers (with all_on_device): the
        CANDIDATE host layers of the online split: a buffer in HBM and one in pinned host memory each, the cache living in one of
        them at a time (host_layers says where it starts) and moved by `move_cache` when the controller changes the host set."""
        sh = model.shape
        self.B, self.smax, self.len = B, smax, 0
        self.serial, self.version = next(_KV_SERIAL), 0      # (serial, version) names one set of kv[] entries: cached pointer tables key on it
        self.pending = {}          # layer -> ticket of a deferred K/V delivery (scheduler.forward, lia_kv_deliver)
        self.all_on_device = all_on_device
        host_layers = frozenset(host_layers) if all_on_device else frozenset()
        dual_layers = frozenset(dual_layers) if all_on_device else frozenset()
        if all_on_device:
            n_gpu = sh.layers          # policy 3 for streamed layers too: every cache lives in HBM
        from . import hostinfo
        shape = (smax, B, sh.heads, sh.head_dim)
        nbytes = 2 * smax * B * sh.heads * sh.head_dim
        # what this generation must ALLOCATE: its host blocks minus the idle ones of the same size an earlier generation left in the
        # pool (r06: the check counted them twice -- a second generate() of a 168 GiB cache was refused on a box that holds it once)
        n_host_blocks = 2 * (sh.layers - n_gpu + len(host_layers | dual_layers))
        # (0.92 of what the container has left, not the planner's 0.85: the caches are the LAST large host allocation of a generation
        # -- the weights are placed by now -- and PinnedPool.acquire guards every block again at 0.93 of the limit; the reference's own
        # large-batch lines sit this close to their box's memory: cxl_offloading.sh:37, batch 1150 = 236 GiB of caches on a 300 GiB box)
        hostinfo.check_host_allocation(max(0, n_host_blocks - PinnedPool.idle_blocks(nbytes)) * nbytes, "host KV cache", safety=0.92)
        self.tensors, self.kv, self._pinned, self.dual = [], [], [], {}

        def device_pair():
            k = torch.empty(shape, dtype=torch.bfloat16, device="cuda")
            return k, torch.empty_like(k)

        def host_pair():
            pk, pv = PinnedPool.acquire(nbytes), PinnedPool.acquire(nbytes)
            self._pinned += [(pk, nbytes), (pv, nbytes)]
            return PinnedPool.as_tensor(pk, shape), PinnedPool.as_tensor(pv, shape)

        for i in range(sh.layers):
            on_dev = 1 if (i < n_gpu and i not in host_layers) else 0
            if i in dual_layers:
                self.dual[i] = {1: device_pair(), 0: host_pair()}
                k, v = self.dual[i][on_dev]
            else:
                k, v = device_pair() if on_dev else host_pair()
            self.tensors.append((k, v))
            self.kv.append(N.KV(k.data_ptr(), v.data_ptr(), smax, B, on_dev))

    def move_cache(self, lib, i, to_device):
        """The cache of dual layer i changes sides: rows [0, len) (contiguous in the seq-major layout) copied over the host link,
        `kv[i]` re-pointed.  Synchronous; the caller has synchronized the compute stream and awaited the layer's K/V delivery."""
        to_device = 1 if to_device else 0
        if i not in self.dual:
            raise ValueError(f"layer {i} has one KV cache only (KVState(dual_layers=...) names the layers that can change sides)")
        if self.kv[i].on_device == to_device:
            return 0
        nbytes = self.len * self.B * self.tensors[i][0].shape[2] * self.tensors[i][0].shape[3] * 2
        dst = self.dual[i][to_device]
        for s_t, d_t in zip(self.tensors[i], dst):
            if nbytes:
                fn = lib.lia_memcpy_h2d if to_device else lib.lia_memcpy_d2h
                N.check(fn(ctypes.c_void_p(d_t.data_ptr()), ctypes.c_void_p(s_t.data_ptr()), ctypes.c_size_t(nbytes)), "lia_memcpy (KV cache move)")
        self.tensors[i] = dst
        self.kv[i] = N.KV(dst[0].data_ptr(), dst[1].data_ptr(), self.smax, self.B, to_device)
        self.version += 1
        return 2 * nbytes

    def close(self):
        """Hand the pinned blocks back to the pool (the tensors over them must not be used afterwards)."""
        if self._pinned and torch.cuda.is_available():
            torch.cuda.synchronize()      # policy-0 K/V deliveries may still be landing in these blocks
        self.tensors, self.kv, self.dual = [], [], {}
        self.version += 1
        for ptr, nbytes in self._pinned:
            PinnedPool.release(ptr, nbytes)
        self._pinned = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class CoopStore:
    """The converged host-layer count of a (model, batch, prompt bucket, new tokens, placement, host tier, wire format, host team)
    configuration, kept in a small JSON file -- ONLY where the user named a place for it (LIA_STATE_DIR; r04 wrote to
    ~/.cache/lia_amd unasked, so a benchmark's cooperative legs depended on whatever ran on the box before, and GPU tests wrote into
    the user's home: ADVICE r04).  The next process with the same key starts its search ON that count -- one stride-3 probe on the
    side the link points to, then +-1 -- instead of at the planner's estimate.  Best effort: without LIA_STATE_DIR, or with an
    unreadable / unwritable file, the search starts from the planner's seed and nothing is written."""

    FILE = "coop_counts.json"

    @staticmethod
    def path():
        d = os.environ.get("LIA_STATE_DIR")
        return os.path.join(d, CoopStore.FILE) if d else None

    @staticmethod
    def load(key):
        if key is None or CoopStore.path() is None:
            return None
        try:
            import json
            with open(CoopStore.path()) as f:
                e = json.load(f).get(key)
            return (int(e["count"]), float(e["ms"])) if e else None
        except (OSError, ValueError, KeyError, TypeError):
            return None

    @staticmethod
    def save(key, count, ms):
        if key is None or ms is None or CoopStore.path() is None:
            return False
        try:
            import fcntl
            import json
            p = CoopStore.path()
            os.makedirs(os.path.dirname(p), exist_ok=True)
            with open(p + ".lock", "w") as lk:        # read-modify-write under an exclusive lock: two processes keep each other's entries
                fcntl.flock(lk, fcntl.LOCK_EX)
                try:
                    with open(p) as f:
                        all_ = json.load(f)
                except (OSError, ValueError):
                    all_ = {}
                all_[key] = {"count": int(count), "ms": round(float(ms), 3)}
                tmp = f"{p}.{os.getpid()}.tmp"
                with open(tmp, "w") as f:
                    json.dump(all_, f, indent=1, sort_keys=True)
                os.replace(tmp, p)                    # whole-file swap: a reader never sees half a file
            return True
        except OSError:
            return False


class CoopController:
    """Online choice of the number of host-computed decode layers (`cpu_layers=-1`, build-defined cooperative split).

    r02 picked the count once from a 4-second calibration (planner.plan_cpu_layers) and landed 129 ... 177 tok/s depending on
    which socket the GPU hangs off.  Here the count follows the MEASURED decode steps: the host set of size c is the first c
    layers of a fixed nested order (`order`, every prefix spread over the streamed layers, raw host copies kept for the first
    c_max only).  A step is max(link time of the streamed layers, host + GPU time): roughly a V in c, but on a shared box the
    bottom is flat to ~1 % with +10 % spikes on single steps, and the first version (a +-1 hill climb on the last sample) walked
    away from the minimum on a spike and stalled two counts below it (results/r03_final2_*).  So this is a pattern search:
    measure the centre, then centre +- stride for stride 3, then 1 -- first on the side the copy engine's idle share points to (link
    >= 97 % busy: more host layers shorten the link time; less: the host is the bottleneck) -- moving the centre whenever a
    candidate beats it by > 0.5 % and carrying on in that direction, twice as far per move after two moves that paid (a seed far
    off; the planner's seed is usually 0-3 counts off, and strides 4 / 2 / 1 spent two steps at a count 40 % slower plus 2 GB of
    cache moves on the first probe: 3.2-3.9 % of a 31-step run in the synthetic box of tests/test_host_logic.py against 2.4-3.2 %); a count's value is the MINIMUM of its last `keep` samples
    (spikes only ever add time).  One settle step after every change (queued copies still reflect the old set).  Converged, it
    stays on the centre -- single slow steps move nothing -- and searches again (+-1 first) after `expire` steps or when six
    steps in a row run > 5 % above the value it converged on (the box changed); every `probe_every` steps it
    re-measures centre +- 1 (a spike on a candidate's single sample can end the search one or two counts off; the second look
    costs five steps at a neighbouring count and repairs that)."""

    STRIDES = (3, 1)
    UP_MIN_BUSY = 0.9             # below this copy-engine busy share no count ABOVE the centre is probed

    def __init__(self, order, start, c_max, expire=96, keep=3, probe_every=16):
        self.order, self.c_max = list(order), min(int(c_max), len(order))
        self.c = max(0, min(int(start), self.c_max))
        self.expire, self.keep, self.probe_every = expire, keep, probe_every
        self.samples = {}             # c -> [(step, ms), ...] the last `keep`
        self.busy_at = {}             # c -> copy-engine busy share of the last sampled step at c
        self.centre, self.stride_i, self.pending, self.direction, self.run = self.c, 0, None, 0, 0
        self.converged_at, self.converged_ms, self.c_conv, self.last_probe, self.probing, self.slow = None, None, None, 0, False, 0
        self.settle, self.step, self.moves, self.searches = 1, 0, 0, 1
        self.bracketed = False        # a minimum has been found once: from then on the search moves by +-1 only
        self.trace = []               # (step, c, ms, link busy share)
        self.store_key, self.seeded = None, False        # (CoopStore) where the converged count is kept across processes

    def host_set(self, c=None):
        return frozenset(self.order[:self.c if c is None else c])

    def superset(self):
        return frozenset(self.order[:self.c_max])

    def restrict(self, c_max):
        """fewer candidates than planned (the container has no room for more raw host copies)"""
        self.c_max = max(0, min(self.c_max, int(c_max)))
        clipped = self.c > self.c_max or self.centre > self.c_max
        self.c, self.centre = min(self.c, self.c_max), min(self.centre, self.c_max)
        self.samples = {k: v for k, v in self.samples.items() if k <= self.c_max}
        if self.pending:
            self.pending = [(c, d) for c, d in self.pending if c <= self.c_max]
        if clipped or self.centre not in self.samples:
            # the point the search stood on is gone (or was never measured): start over from the clipped centre -- a half-finished
            # comparison against a centre without samples would compare with None
            self.c = self.centre
            self.pending, self.direction, self.run, self.stride_i = None, 0, 0, (len(self.STRIDES) - 1 if self.bracketed else 0)
            self.converged_at, self.converged_ms, self.c_conv, self.probing, self.slow = None, None, None, False, 0
            self.settle = max(self.settle, 1)

    def new_sequence(self):
        """a new generation: its first decode step also waits for the prefill's K/V deliveries and loads layers on demand"""
        self.settle = max(self.settle, 1)

    def value(self, c):
        v = [ms for st, ms in self.samples.get(c, ()) if self.step - st <= self.expire]
        return min(v) if v else None

    def _go(self, c):
        if c != self.c:
            self.c, self.settle, self.moves = c, 1, self.moves + 1
        return self.c

    def _candidates(self, busy, again=False):
        s = self.STRIDES[self.stride_i]
        busy = self.busy_at.get(self.centre, busy)      # the copy engine's share AT THE CENTRE (the last step may have run at a candidate)
        side = (1, -1) if busy >= 0.97 else (-1, 1)
        if busy < self.UP_MIN_BUSY:
            side = (-1,)                # the link idles > 10 % of the step: the host side is the bottleneck, one more host layer cannot pay
        return [(self.centre + d * s, d) for d in side
                if 0 <= self.centre + d * s <= self.c_max and (again or self.value(self.centre + d * s) is None)]

    def observe(self, step_ms, busy_share):
        """one finished decode step at the current c -> the c of the next step"""
        self.step += 1
        self.trace.append((self.step, self.c, round(step_ms, 2), round(busy_share, 3)))
        if self.settle > 0:
            self.settle -= 1
            return self.c
        self.samples[self.c] = (self.samples.get(self.c, []) + [(self.step, step_ms)])[-self.keep:]
        self.busy_at[self.c] = busy_share
        if self.converged_at is not None and not self.probing:
            self.slow = self.slow + 1 if step_ms > 1.05 * self.converged_ms else 0
            if self.step - self.converged_at >= self.expire or self.slow >= 2 * self.keep:
                # search again around where we are, +-1 first (a move that pays carries on, twice as far after two): everything
                # measured before is stale.  (Six slow steps in a row, not three: a noise episode of 3-4 steps restarted the
                # search with a stride-3 probe 25 % off the optimum -- results/r03_final10_*.)
                self.samples = {self.c: self.samples[self.c][-1:]}
                self.centre, self.stride_i, self.pending, self.direction, self.run = self.c, len(self.STRIDES) - 1, None, 0, 0
                self.converged_at, self.searches, self.slow = None, self.searches + 1, 0
            elif self.step - self.last_probe >= self.probe_every:
                self.stride_i, self.probing = len(self.STRIDES) - 1, True
                self.pending = self._candidates(busy_share, again=True)
            else:
                return self.c
        if self.pending is None:                                  # the centre has just been measured: open this stride
            self.pending = self._candidates(busy_share)
        elif self.c != self.centre:                               # a candidate has just been measured
            if self.value(self.centre) is None:                   # the centre's samples expired / were dropped: measure it again first
                self.pending = [(self.c, self.direction)] + list(self.pending or [])
                self.direction = 0
                return self._go(self.centre)
            if self.value(self.c) < 0.995 * self.value(self.centre):
                d, s = self.direction, self.STRIDES[self.stride_i]
                self.centre, self.run = self.c, self.run + 1
                far = 2 if (self.run >= 2 and not self.bracketed) else 1
                nxt = self.centre + d * s * far                   # it paid: carry on the same way before looking back
                if d > 0 and busy_share < self.UP_MIN_BUSY:
                    nxt = self.centre
                nxt = max(0, min(self.c_max, nxt))
                self.pending = [(nxt, d)] if nxt != self.centre and self.value(nxt) is None else []
            else:
                self.run = 0
        while not self.pending:
            if self.stride_i + 1 >= len(self.STRIDES):
                if self.converged_at is None or self.centre != self.c_conv:
                    self.converged_at, self.converged_ms, self.c_conv = self.step, self.value(self.centre), self.centre
                    CoopStore.save(self.store_key, self.centre, self.converged_ms)
                self.last_probe, self.probing, self.bracketed = self.step, False, True
                return self._go(self.centre)
            self.stride_i, self.run = self.stride_i + 1, 0
            self.pending = self._candidates(busy_share)
        cand, self.direction = self.pending.pop(0)
        return self._go(cand)

    def report(self):
        return {"host_layers": self.c, "centre": self.centre, "converged": self.converged_at is not None, "max_host_layers": self.c_max,
                "seeded_from_store": self.seeded,
                "moves": self.moves, "searches": self.searches, "steps_observed": self.step,
                "ms_by_count": {str(k): round(self.value(k), 2) for k in sorted(self.samples) if self.value(k) is not None},
                "trace_tail": self.trace[-12:]}


class OffloadScheduler:
    """forward(input_ids, kv_state, **lia flags) -> (logits [B,vocab], next_ids [B]) on the device."""

    def __init__(self, model, device=0, n_slots=None, dp_group=None, wire=None):
        import os
        self.model, self.device, self.dp = model, device, dp_group
        self.n_slots = n_slots or 4
        # wire format of the streamed layers: "pack10" (the lossless encoding of lia_pack10.hip) or "raw" bf16
        fmt = default_stream_format() if wire is None else wire
        self.wire = wire_format_code(fmt)
        self.ctx = None
        self.ws_rows = 0
        self.pipe = None
        self.hidden = {}
        self.resident_ptrs = {}
        self.last_step_ms = {}
        self.host_threads = None    # OpenMP team of the host attention / host layers; default = hostinfo.default_host_threads
        # policy-0 prefill: keep the fresh K/V of the streamed layers in HBM until the prefill's last layer has run and deliver
        # them to the host caches then (lia_kv_deliver), instead of beside the prefill's weight stream.  LIA_DEFER_KV=0: deliver at once.
        self.defer_kv = os.environ.get("LIA_DEFER_KV", "1") != "0"
        # the last layer of a prefill computes everything behind its q|k|v projection on the last position only (lia_layer_forward_last);
        # prefill_tail = False: every position, as the reference does (same ids, tests/test_gpu_round3.py)
        self.prefill_tail = True
        self._kv_hold = None        # ((B, T, first streamed layer), [(k, v, KV struct) per streamed layer])
        # every delivery ticket not yet waited for, whichever KVState it belongs to: ticket -> (weakref(KVState), layer).  The
        # holding caches are shared by all generations of this scheduler, so a new prefill may only write them once ALL of these
        # have landed (a generation that ended at its prefill, a second live KVState, a KVState reused with len reset)
        self._outstanding = {}
        self._coop = None           # CoopController of the cooperative split (cpu_layers=-1), kept across generations
        self._coop_key = None
        self.kv_moved_bytes = 0      # KV cache bytes moved between HBM and host by the online split (KVState.move_cache)
        self.host_layer_time = {"ms": 0.0, "calls": 0}     # wall clock inside lia_host_layer(s)_forward (policy 1 / host-computed layers); run_generation --profile reads and resets it
        self.kv_delivery = {"bytes": 0, "device_ms": None, "host_wait_ms": 0.0}   # of the last deferred delivery

    # -- resources -----------------------------------------------------------------------------------
    def _ensure(self, rows, B, T, n_gpu):
        """Context workspace (sized by the largest layer call seen so far), streamer slots, hidden buffers."""
        sh = self.model.shape
        lm_bytes = 2 * 256 * sh.hidden + 8 * min(B, 256) * sh.vocab * 4 + (1 << 20)
        need = max(ops.workspace_bytes(self.model.desc, rows), lm_bytes)
        if self.ctx is None or need > self.ctx.workspace_bytes:
            if self.pipe is not None:
                self.pipe.close()
                self.pipe = None
            if self.ctx is not None:
                self._await_all_deliveries()      # their tickets die with the context
                self.ctx.close()
            # (the library allocates with hipMalloc: blocks torch's caching allocator keeps from the model's placement -- the drawn
            # layers, the wire encoder's buffers -- are handed back first; at batch 1050 the planner's pick ran out of HBM here)
            torch.cuda.empty_cache()
            self.ctx = ops.Context(self.device, need)
            # the policy-2 host attention team: usable CPUs (affinity mask AND cgroup quota) split over the ranks, never
            # omp_get_max_threads() -- a team larger than the quota stalls every layer's round trip (hostinfo.py)
            from . import hostinfo
            if not getattr(self, "host_threads", None):
                self.host_threads = hostinfo.default_host_threads(self.dp.world if self.dp else 1)
            self.ctx.set_host_threads(self.host_threads)
            hostinfo.cap_torch_threads(self.host_threads)     # torch's own CPU ops of the token loop obey the same bound
        if self.pipe is None and n_gpu < sh.layers:
            torch.cuda.empty_cache()
            self.pipe = WeightPipeline(self.ctx, self.model, self.n_slots, self.dp)
        key = (B, T)
        if key not in self.hidden:
            if len(self.hidden) > 4:
                self.hidden.clear()
            self.hidden[key] = (torch.empty((B, T, sh.hidden), dtype=torch.bfloat16, device="cuda"),
                                torch.empty((B, T, sh.hidden), dtype=torch.bfloat16, device="cuda"))
        return self.hidden[key]

    def _hold_caches(self, B, T, n_gpu):
        """Device holding caches [T, B, h, d] x 2 for every streamed layer (20.7 GB for OPT-30B at B = 64, T = 256), or None when
        HBM has no room for them (the deliveries then run beside the prefill as in r01)."""
        sh = self.model.shape
        key = (B, T, n_gpu)
        if self._kv_hold is not None and self._kv_hold[0] == key:
            return self._kv_hold[1]
        self._kv_hold = None
        need = 2 * (sh.layers - n_gpu) * T * B * sh.hidden * 2
        free, _ = torch.cuda.mem_get_info()
        if need > 0.5 * free:
            return None
        out = []
        for _ in range(n_gpu, sh.layers):
            k = torch.empty((T, B, sh.heads, sh.head_dim), dtype=torch.bfloat16, device="cuda")
            v = torch.empty_like(k)
            out.append((k, v, N.KV(k.data_ptr(), v.data_ptr(), T, B, 1)))
        self._kv_hold = (key, out)
        return out

    def _await_kv(self, kv_state, idx=None):
        """block until the deferred K/V delivery of layer idx (None: of every layer) has reached the host cache"""
        pend = getattr(kv_state, "pending", None)
        if not pend:
            return
        import time
        t0 = time.time()
        for i in ([idx] if idx is not None else sorted(pend)):
            t = pend.pop(i, None)
            if t is not None:
                N.check(self.ctx.lib.lia_kv_deliver_wait(self.ctx.handle, t), "lia_kv_deliver_wait")
                self._outstanding.pop(t, None)
        self.kv_delivery["host_wait_ms"] += 1e3 * (time.time() - t0)
        if not pend and self.kv_delivery.get("_issued_at") is not None:
            self.kv_delivery.pop("_issued_at")
            ms = ctypes.c_double()
            N.check(self.ctx.lib.lia_kv_deliver_batch_ms(self.ctx.handle, ctypes.byref(ms)), "lia_kv_deliver_batch_ms")
            self.kv_delivery["device_ms"] = ms.value          # first copy's start -> last copy's end on the delivery stream

    def _await_all_deliveries(self):
        """every outstanding delivery of this scheduler, whichever generation issued it: their copies read the shared holding
        caches on the D2H stream, which the compute stream is about to overwrite"""
        for t, (ref, idx) in list(self._outstanding.items()):
            N.check(self.ctx.lib.lia_kv_deliver_wait(self.ctx.handle, t), "lia_kv_deliver_wait")
            owner = ref()
            if owner is not None and getattr(owner, "pending", None):
                owner.pending.pop(idx, None)
        if self._outstanding:
            ms = ctypes.c_double()
            N.check(self.ctx.lib.lia_kv_deliver_batch_ms(self.ctx.handle, ctypes.byref(ms)), "lia_kv_deliver_batch_ms")   # close the timing batch
            self.kv_delivery.pop("_issued_at", None)
        self._outstanding.clear()

    def _resident(self, idx):
        if idx not in self.resident_ptrs:
            self.resident_ptrs[idx] = ops.weight_ptr_array(self.model.layers[idx].device_ptr(), self.model.offsets)
        return self.resident_ptrs[idx]

    # -- one forward -----------------------------------------------------------------------------------
    # forward = _validate (shapes and flags -> a step record) -> _place (host set, cache sides, layer tiers) -> embed ->
    # _run_layers (resident run, streamed / host-computed layers) -> lm_head -> _deliver (K/V to the host caches, synchronize,
    # the cooperative controller's sample).  Each phase reads and extends the step record `s`; none reaches back.
    def forward(self, input_ids, kv_state, prefill_policy=1, decoding_policy=1, no_overlap=False, pin_weight=False,
                gpu_percentage=0, num_minibatch=1, enable_cxl=False, max_new_tokens=None, suppress_token=-1, cpu_layers=0,
                cpu_layers_start=None):
        """cpu_layers (build-defined): with decoding policy 2, that many of the streamed layers run
        their decode step entirely on the host cores (policy 1, weights read in place) instead of crossing the link --
        the cooperative split of the reference taken per layer.  Prefill is unaffected (policy 0 for every layer).
        cpu_layers=-1: the count is chosen ONLINE (CoopController) from the measured decode steps, starting at cpu_layers_start
        (default: planner.plan_cpu_layers for this shape and box defaults)."""
        s = self._validate(input_ids, kv_state, prefill_policy, decoding_policy, no_overlap, gpu_percentage, num_minibatch)
        self._place(s, kv_state, pin_weight, enable_cxl, max_new_tokens, cpu_layers, cpu_layers_start)
        m, sh, ctx = self.model, self.model.shape, self.ctx
        ids_dev = input_ids.to("cuda", non_blocking=False).contiguous()
        N.check(ctx.lib.lia_embed(ctypes.c_void_p(ids_dev.data_ptr()), ctypes.c_void_p(m.embed_tokens.data_ptr()),
                                  ctypes.c_void_p(m.embed_positions.data_ptr()), ctypes.c_void_p(s.x.data_ptr()), s.B, s.T, s.pos0,
                                  sh.hidden, ctypes.c_void_p(ctx.stream)), "lia_embed")
        if s.policy == 1 and s.n_gpu < s.L:
            self._await_kv(kv_state)
            hidden = self._host_layers(s.x, kv_state, s.n_gpu, s.B, s.T, s.pos0)  # resident prefix on the GPU, the rest on the CPU
            logits, nxt = ctx.lm_head(hidden, m.final_ln_w, m.final_ln_b, m.embed_tokens, sh.ln_eps, suppress_token)
            ctx.synchronize()
            kv_state.len = s.pos0 + s.T
            return logits, nxt
        hidden = self._run_layers(s, kv_state)
        logits, nxt = ctx.lm_head(hidden, m.final_ln_w, m.final_ln_b, m.embed_tokens, sh.ln_eps, suppress_token)
        self._deliver(s, kv_state)
        return logits, nxt

    def _validate(self, input_ids, kv_state, prefill_policy, decoding_policy, no_overlap, gpu_percentage, num_minibatch):
        """shapes and flags of one forward -> the step record; raises on the combinations the path does not have"""
        from types import SimpleNamespace
        sh = self.model.shape
        B, T = input_ids.shape
        L = sh.layers
        n_gpu = int(L * gpu_percentage / 100)                      # lia/modeling_opt.py:1182
        is_prefill = T != 1                                        # :1186-1188
        policy = prefill_policy if is_prefill else decoding_policy
        if n_gpu < L and policy not in (0, 1, 2, 3):
            raise ValueError(f"unsupported policy {policy} (prefill: 0, 1 or 3; decode: 0, 1, 2 or 3)")
        if n_gpu < L and (policy == 3) != bool(kv_state.all_on_device):
            raise ValueError("policy 3 on streamed layers (KV cache in HBM, SURVEY.md section 8 f-1) must be chosen for BOTH "
                             "phases: the cache of a generation lives either in HBM or on the host")
        if is_prefill and policy == 2 and n_gpu < L:
            raise ValueError("prefill policy must be 0 on the GPU path (the reference has no prefill-2 branch)")
        # :1178 mini_bsz = int(bsz / num_minibatch), looped num_minibatch times: the reference leaves the rows behind the last full
        # minibatch unwritten when the division has a remainder and computes nothing at all for num_minibatch > bsz -- both happen in
        # its own scripts (cxl_offloading.sh:37 `--batch-size 1150 --num-minibatch 3`, lia_offline.sh:27-29 `--batch-size 1
        # --num-minibatch 2`).  Here: the intended semantics -- num_minibatch clamped to the batch, the last
        # minibatch takes the remainder (minibatch_bounds)
        minis = minibatch_bounds(B, num_minibatch) if (policy in (0, 3) and is_prefill) or policy == 0 else [(0, B)]
        mini = max(n for _, n in minis)
        return SimpleNamespace(B=B, T=T, L=L, n_gpu=n_gpu, is_prefill=is_prefill, policy=policy, prefill_policy=prefill_policy,
                               decoding_policy=decoding_policy, gpu_percentage=gpu_percentage, mini=mini, minis=minis, overlap=not no_overlap,
                               pos0=kv_state.len, coop=None, cpu_set=frozenset(), host_act=frozenset(), host_now=frozenset(),
                               t_fwd0=None, busy0=0.0, hold=None, x=None, y=None)

    def _place(self, s, kv_state, pin_weight, enable_cxl, max_new_tokens, cpu_layers, cpu_layers_start):
        """the host-computed layer set (fixed or the online controller's), the side every cache lives on, the layers' tiers
        (move_gpu_layer / pin_memory, idempotent), the hidden-state buffers and the holding caches of a deferred K/V delivery"""
        m = self.model
        n_gpu, L, B, T, is_prefill, decoding_policy = s.n_gpu, s.L, s.B, s.T, s.is_prefill, s.decoding_policy
        # The policy-1 host path reads the host copy directly, so the packed wire format is only used when neither phase runs on the CPU.
        coop = None
        if cpu_layers and cpu_layers < 0 and decoding_policy in (2, 3) and self.dp is None and n_gpu < L - 1:
            coop = self._coop_controller(n_gpu, L, B, T, max_new_tokens, s.gpu_percentage, decoding_policy, cpu_layers_start)
            self._fit_host_candidates(coop, enable_cxl and pin_weight)
            cpu_set = coop.superset()                                # layers that keep a raw host copy (and a host KV cache)
        else:
            cpu_set = self.cpu_layer_set(n_gpu, L, cpu_layers) if (cpu_layers and cpu_layers > 0 and decoding_policy in (2, 3) and self.dp is None) else frozenset()
        if cpu_set and decoding_policy == 3:
            # the host-computed layers need their cache on the host; with the online count a candidate's cache follows the host set
            need_host = coop.host_set() if coop is not None else cpu_set
            fixed = [i for i in need_host if i not in getattr(kv_state, "dual", {}) and kv_state.kv[i].on_device]
            if fixed:
                raise ValueError("cpu_layers with the KV cache in HBM: the host-computed layers need a host cache "
                                 "(KVState(..., all_on_device=True, host_layers=OffloadScheduler.cpu_layer_set(...)), "
                                 f"dual_layers=... for the online count); layers {fixed[:4]}")
            if is_prefill and kv_state.len == 0:
                for i in getattr(kv_state, "dual", {}):                  # an empty cache changes sides for free
                    kv_state.move_cache(N.lib(), i, to_device=(i not in need_host))
        shard = (self.dp.rank, self.dp.world) if (self.dp is not None and self.dp.world > 1 and self.dp.mode == "allgather") else None
        wire, raw_set = placement_formats(s.prefill_policy, decoding_policy, self.wire, n_gpu, L, pin_weight, enable_cxl, cpu_set, self.dp is not None)
        if m.placed_for != m._place_key(n_gpu, pin_weight, enable_cxl, wire, raw_set, shard):
            # the flags changed since the last placement: the model re-tiers its layers (policy 1 wants raw host copies,
            # another gpu%, wire format or host tier).  Copies in flight read host buffers that are about to be freed and
            # the cached device pointers of the resident layers go stale.
            if self.pipe is not None:
                self.pipe.drain()
            self.resident_ptrs.clear()
            self._run_key = None
        m.place(n_gpu, pin_weight, enable_cxl, wire, raw_layers=raw_set, shard=shard)
        if coop is not None and is_prefill:
            coop.new_sequence()                                      # the first decode step is not a sample of the steady state
        s.coop, s.cpu_set = coop, cpu_set
        s.host_act = coop.host_set() if coop is not None else cpu_set   # the layers whose DECODE step runs on the host cores
        s.host_now = s.host_act if not is_prefill else frozenset()    # layers this forward computes on the host
        if coop is not None and not is_prefill:
            import time
            s.t_fwd0, s.busy0 = time.time(), (self.pipe.poll_stats()[1] if self.pipe else 0.0)
        rows = B * T if n_gpu > 0 else s.mini * T                  # resident layers take the whole batch
        if s.decoding_policy == 0 and n_gpu < L:
            # policy-0 DECODE parks the cached prefix in a slab.  (Until r06 the test was on the PHASE's policy, so every policy-0
            # prefill sized the workspace for batch x max positions rows: 60 GB of HBM at batch 1050 x 288 positions, which is
            # what ran the planner's batch-1050 pick out of memory; the reference's scripts never use decode policy 0.)
            rows = max(rows, s.mini * kv_state.smax)
        s.x, s.y = self._ensure(rows, B, T, n_gpu)
        if is_prefill and s.policy == 0 and n_gpu < L and self.defer_kv and s.pos0 == 0:
            self._await_all_deliveries()          # (host-blocking: afterwards nothing on the D2H stream reads the holding caches)
            s.hold = self._hold_caches(B, T, n_gpu)

    def _run_layers(self, s, kv_state):
        """the decoder layers of one forward on the GPU path: the resident run, then every streamed layer behind its weight copy
        (or on the host cores, for the cooperative split's host set); returns the hidden state that feeds lm_head"""
        m, sh, ctx, pipe = self.model, self.model.shape, self.ctx, self.pipe
        B, T, L, n_gpu, is_prefill, policy, mini, overlap, pos0, hold = s.B, s.T, s.L, s.n_gpu, s.is_prefill, s.policy, s.mini, s.overlap, s.pos0, s.hold
        host_act, host_now, x, y = s.host_act, s.host_now, s.x, s.y
        first_streamed = n_gpu

        def next_streamed(i, wrapped):
            """the streamed layer after i that needs a slot: decode forwards skip the host-computed layers"""
            while True:
                i += 1
                if i >= L:
                    i, wrapped = first_streamed, True
                if not ((wrapped or not is_prefill) and i in host_act):
                    return i, wrapped

        if n_gpu < L and overlap:
            pipe.prefetch(first_streamed)                          # no-op if the previous step already wrapped to it
        # the prefill's last layer: only hidden[:, -1, :] feeds lm_head (models.py:424-431), its K/V of every position feed decode
        tail_last = self.prefill_tail and is_prefill and T > 1 and pos0 == 0 and (policy in (0, 3) or n_gpu == L) and (L - 1) not in host_now
        xlast = None
        first = 0
        if not is_prefill and n_gpu > 0:
            # decode: the resident run (whole batch, everything on the GPU incl. KV -- policy 3; :1246-1260) in ONE library call
            # (lia_decode_layers: the per-op route layer by layer, or with LIA_FUSED_DECODE=1 an attention launch and a
            # persistent chain launch per layer, lia_chain.hip)
            key = (n_gpu, kv_state.serial, kv_state.version, self._resident(0)[0])
            if getattr(self, "_run_key", None) != key:
                ptrs = []
                for i in range(n_gpu):
                    ptrs.extend(self._resident(i))
                self._run = ((ctypes.c_void_p * (16 * n_gpu))(*ptrs), (ctypes.POINTER(N.KV) * n_gpu)(*[ctypes.pointer(kv_state.kv[i]) for i in range(n_gpu)]))
                self._run_key = key
            N.check(ctx.lib.lia_decode_layers(ctx.handle, ctypes.byref(m.desc), n_gpu, ctypes.cast(self._run[0], ctypes.c_void_p),
                                              ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(y.data_ptr()), ctypes.cast(self._run[1], ctypes.c_void_p),
                                              B, pos0, ctypes.c_void_p(ctx.stream)), "lia_decode_layers")
            x, y = y, x
            first = n_gpu
        for idx in range(first, L):
            if idx < n_gpu:
                # resident layer: whole batch, everything on the GPU incl. KV (policy 3; :1246-1260)
                if tail_last and idx == L - 1:
                    xlast = torch.empty((B, 1, sh.hidden), dtype=torch.bfloat16, device="cuda")
                    ctx.layer_forward_last(m.desc, 3, self._resident(idx), x, xlast, kv_state.kv[idx], B, T, pos0, 0)
                    continue
                ctx.layer_forward(m.desc, 3, self._resident(idx), x, y, kv_state.kv[idx], B, T, pos0, 0)
                x, y = y, x
                continue
            if hold is None:
                self._await_kv(kv_state, idx)                      # a deferred K/V delivery of an earlier prefill: the host cache must be complete
            if idx in host_now:
                x, y = self._host_decode_layer(idx, x, y, kv_state, B, T, pos0)
                continue
            if not overlap:
                pipe.prefetch(idx)
            wptrs = pipe.acquire(idx)
            if overlap:
                # weight prefetch of the next streamed layer(s) overlaps this layer's compute (:1305-1312, :1508-1515):
                # every free slot is filled; after the last layer the order wraps to the first streamed layer of the
                # NEXT forward, so the copy engine keeps running across token steps
                nxt, wrapped = idx, False
                for _ in range(pipe.n_slots - 1):
                    nxt, wrapped = next_streamed(nxt, wrapped)
                    if nxt == idx or not pipe.can_prefetch():
                        break
                    pipe.prefetch(nxt)
            if policy in (0, 3):
                # FlexGen-style minibatches (:1283-1365).  Policy 3 here = streamed weights with the cache kept in HBM
                # (build-defined; the reference reserves 3 for resident layers, :1175-1176)
                # a host-computed layer keeps its cache on the host: its prefill delivers K/V there (policy 0) even when the
                # other streamed layers keep theirs in HBM (policy 3)
                kvl = kv_state.kv[idx]
                pol = 0 if (policy == 3 and not kvl.on_device) else policy
                if pol == 0 and not is_prefill and policy == 3:
                    pol = 2            # a host layer of the split that the GPU computes this step: its cache lives on the host
                if hold is not None and pol == 0:
                    pol, kvl = 3, hold[idx - n_gpu][2]             # same arithmetic, rows land in the HBM holding cache
                if tail_last and idx == L - 1:
                    xlast = torch.empty((B, 1, sh.hidden), dtype=torch.bfloat16, device="cuda")
                for b0, nb in s.minis:
                    sl = slice(b0, b0 + nb)
                    if tail_last and idx == L - 1:
                        ctx.layer_forward_last(m.desc, pol, wptrs, x[sl], xlast[sl], kvl, nb, T, pos0, b0)
                    else:
                        ctx.layer_forward(m.desc, pol, wptrs, x[sl], y[sl], kvl, nb, T, pos0, b0)
            else:
                ctx.layer_forward(m.desc, 2, wptrs, x, y, kv_state.kv[idx], B, T, pos0, 0)   # :1493-1543
            pipe.release(idx)
            x, y = y, x
            if not overlap:
                ctx.synchronize()
        return x if xlast is None else xlast

    def _deliver(self, s, kv_state):
        """behind lm_head: the deferred K/V deliveries of a policy-0 prefill (tickets, awaited per layer by the first decode
        step), the synchronisation the policy needs, the new cache length, and the cooperative controller's sample of the step"""
        sh, ctx, pipe = self.model.shape, self.ctx, self.pipe
        B, T, L, n_gpu, hold = s.B, s.T, s.L, s.n_gpu, s.hold
        if hold is not None:
            # the deferred deliveries, in the order the first decode step will need them; tickets are awaited per layer there
            import time
            import weakref
            self._await_kv(kv_state)              # (a KVState reused for a second prefill: never drop a ticket unawaited)
            kv_state.pending = {}
            self.kv_delivery = {"bytes": 0, "device_ms": None, "host_wait_ms": 0.0, "_issued_at": time.time()}
            for idx in range(n_gpu, L):
                if kv_state.kv[idx].on_device:
                    continue
                t = ctypes.c_int()
                N.check(ctx.lib.lia_kv_deliver(ctx.handle, ctypes.byref(hold[idx - n_gpu][2]), ctypes.byref(kv_state.kv[idx]), T,
                                               sh.hidden, ctypes.byref(t)), "lia_kv_deliver")
                kv_state.pending[idx] = t.value
                self._outstanding[t.value] = (weakref.ref(kv_state), idx)
                self.kv_delivery["bytes"] += 2 * T * B * sh.hidden * 2
            N.check(ctx.lib.lia_ctx_synchronize_compute(ctx.handle), "lia_ctx_synchronize_compute")
        else:
            ctx.synchronize()
            if s.policy == 0 or (s.is_prefill and s.cpu_set):
                ctx.kv_store_wait()                                # host cache complete before the next step reads it
        kv_state.len = s.pos0 + T
        coop = s.coop
        if s.t_fwd0 is not None:
            import time
            step_ms = 1e3 * (time.time() - s.t_fwd0)
            busy = (self.pipe.poll_stats()[1] - s.busy0) / max(step_ms, 1e-6) if self.pipe else 0.0
            before = coop.host_set()
            coop.observe(step_ms, min(busy, 1.0))
            after = coop.host_set()
            for li in after - before:                             # newly host-computed: a queued copy of it will never be used
                pipe.forget(li)
            if s.decoding_policy == 3 and after != before:
                # KV in HBM: the cache of a layer that changes sides follows it (0.5 GB per layer at the headline shape, ~10 ms)
                for li in sorted(after ^ before):
                    self._await_kv(kv_state, li)
                    self.kv_moved_bytes += kv_state.move_cache(ctx.lib, li, to_device=(li not in after))

    def _coop_controller(self, n_gpu, L, B, T, max_new_tokens, gpu_percentage, decoding_policy, start):
        key = (n_gpu, L, B, decoding_policy)
        if self._coop is None or self._coop_key != key:
            seeded_ok = start is None            # an explicit start (tests, LIA_COOP_START) is taken as given
            if start is None:
                from . import hostinfo, planner
                start, _ = planner.plan_cpu_layers(self.model.shape, B, T, max_new_tokens or 32, gpu_percentage,
                                                   planner.Box(host_threads=self.host_threads or hostinfo.default_host_threads(1),
                                                               wire_ratio={0: 1.0, 10: 0.675}[self.wire]),
                                                   kv_in_hbm=(decoding_policy == 3))
            order = self.cpu_layer_order(n_gpu, L)
            sh = self.model.shape
            from . import hostinfo
            # everything that moves the optimum is in the key: the prompt length in buckets of a power of two (the host attention and
            # the caches' size scale with it), the new tokens, which tier the streamed layers live in
            tiers = sorted({st.tier for st in getattr(self.model, "layers", [])[n_gpu:] if getattr(st, "tier", None) not in ("device", None)}) or ["-"]
            t_bucket = 1 << max(0, int(T) - 1).bit_length()
            store_key = "|".join(str(v) for v in (sh.name, sh.hidden, sh.ffn, L, n_gpu, B, f"T<={t_bucket}", f"new{max_new_tokens or 0}", decoding_policy,
                                                  self.wire, "+".join(tiers), self.host_threads or hostinfo.default_host_threads(1)))
            kept = CoopStore.load(store_key) if seeded_ok else None
            if kept is not None:
                start = max(0, min(kept[0], len(order)))
            self._coop = CoopController(order, start, min(len(order), int(start) + COOP_HEADROOM))
            self._coop.store_key = store_key
            if kept is not None:
                # a count this box converged on before for exactly this configuration: the search starts ON it, but still opens with
                # the stride-3 probe (one candidate, on the side the link points to) before the +-1 round -- a stored count that is
                # wrong for today's box is left in two moves, not walked away from one count at a time
                self._coop.seeded = True
            self._coop_key = key
        return self._coop

    def _fit_host_candidates(self, coop, in_numa_tier, extra_per_layer=0, trim_pool=False):
        """Shrink the controller's candidate set to the raw host copies the container has room for (OPT-175B at gpu% = 5 in a
        300 GiB container: the planner's count + 10 does not fit, and the per-allocation guard would refuse the placement halfway
        through, in layer-index order -- a clustered host set).  A candidate in the NUMA tier swaps its packed copy for the raw
        one (growth = the difference, guard ceiling 0.93); a pinned one keeps both (growth = the raw bytes, ceiling 0.85).
        extra_per_layer: what else a candidate will pin (its host KV buffer when the caches otherwise live in HBM)."""
        layers = self.model.layers
        need = [k for k, li in enumerate(coop.order[:coop.c_max]) if layers[li].raw_host_ptr() is None and layers[li].tier != "remote"]
        if not need:
            return                                   # (every decode step comes through here: nothing to read once the copies exist)
        if trim_pool:
            PinnedPool.trim()                        # idle cache blocks of earlier generations count against the room
        from . import hostinfo
        mem = hostinfo.cgroup_memory()
        if mem["max"] is None or mem["current"] is None:
            return
        st = layers[coop.order[need[0]]]
        grow = ((st.nbytes - (st.stream_bytes if st.packed else 0)) if in_numa_tier else st.nbytes) + int(extra_per_layer)
        room = (0.93 if in_numa_tier else 0.85) * mem["max"] - mem["current"] - st.nbytes          # (one layer of slack: the guard looks at the transient)
        fit = max(0, int(room // max(grow, 1)))
        if fit < len(need):
            new_max = need[fit]                      # candidates before the first one that does not fit
            if new_max <= 0:
                # cpu_layers = -1 asks for "as many as pay": with no room for even one raw copy that is zero host layers -- the plain
                # streamed configuration -- not an aborted generation (a fixed cpu_layers > 0 never comes through here; its
                # placement fails in the allocation guard, loudly)
                import warnings
                warnings.warn(f"cooperative split: no room for a raw host copy of one layer ({st.nbytes / 2**30:.2f} GiB) in this "
                              "container; running with zero host-computed layers", RuntimeWarning, stacklevel=2)
            coop.restrict(max(new_max, 0))

    def host_team_report(self):
        g = getattr(self, "_host_gov", None)
        return None if g is None else {"threads": g.threads}

    def coop_report(self):
        return self._coop.report() if self._coop is not None else None

    @staticmethod
    def cpu_layer_order(n_gpu, L):
        """A fixed NESTED order of the streamed layers (never the first one, which the wrap-around prefetch targets) whose every
        prefix is spread over the layer range: the candidates sorted by the bit-reversal of their position (van der Corput), so the
        host set can grow or shrink by one layer without moving the others and never holds runs of neighbours while it is small
        enough not to.  (A greedy farthest-point order looks as even for the first m / 3 picks and then fills one end of the range
        with neighbours: eight host layers in a row leave the copy engine idle, 426 vs 404 ms per step at 17 layers -- r03 results.)"""
        cand = list(range(n_gpu + 1, L))
        bits = max(1, (len(cand) - 1).bit_length())
        rev = lambda i: int(format(i, f"0{bits}b")[::-1], 2)      # noqa: E731
        return [cand[i] for i in sorted(range(len(cand)), key=rev)]

    @staticmethod
    def cpu_layer_set(n_gpu, L, count):
        """`count` host-computed layers spread evenly over the streamed ones (never the first streamed layer, which the
        wrap-around prefetch targets): while the host works on one of them the copy engine fills the free slots."""
        n_str = L - n_gpu
        count = max(0, min(int(count), n_str - 1))
        if count == 0:
            return frozenset()
        return frozenset(n_gpu + 1 + int((j + 0.5) * (n_str - 1) / count) for j in range(count))

    def _host_team(self, world=1):
        """the whole-layer host team (hostinfo.HostTeam): the attention team's count"""
        if getattr(self, "_host_gov", None) is None:
            from . import hostinfo
            self._host_gov = hostinfo.HostTeam(getattr(self, "host_threads", None) or hostinfo.default_host_threads(world))
        return self._host_gov

    def _hidden_pair(self, nbytes):
        """Two pinned hidden-state buffers for the host-computed layers, kept across steps; a larger request hands the old
        pair back to the pool first, close() hands back the last one."""
        cur = getattr(self, "_host_hidden", None)
        if cur is None or cur[2] < nbytes:
            self._release_hidden_pair()
            self._host_hidden = cur = (PinnedPool.acquire(nbytes), PinnedPool.acquire(nbytes), nbytes)
        return cur[0], cur[1]

    def _release_hidden_pair(self):
        cur = getattr(self, "_host_hidden", None)
        if cur is not None:
            self.ctx.synchronize()                 # a blit into / out of the pair may still be in flight
            PinnedPool.release(cur[0], cur[2])
            PinnedPool.release(cur[1], cur[2])
            self._host_hidden = None

    def _host_decode_layer(self, idx, x, y, kv_state, B, T, pos0):
        """One decode step of layer idx on the host cores (policy 1 for this layer): hidden state GPU -> pinned host by a
        kernel blit, lia_host_layer_forward on the raw host copy of the weights and the host KV cache, result back."""
        m, sh, ctx, lib = self.model, self.model.shape, self.ctx, self.ctx.lib
        st = m.layers[idx]
        raw = st.raw_host_ptr()
        if raw is None:
            raise ValueError(f"layer {idx} is to run on the host but has no raw bf16 host copy")
        nbytes = B * T * sh.hidden * 2
        hx, hy = self._hidden_pair(nbytes)
        N.check(lib.lia_blit(ctypes.c_void_p(hx), ctypes.c_void_p(x.data_ptr()), nbytes, ctypes.c_void_p(ctx.stream)), "lia_blit")
        ctx.synchronize()
        from . import hostinfo
        threads = self._host_team(1).threads
        w = ops.weight_ptr_array(raw, m.offsets)
        kv = kv_state.kv[idx]
        import time
        t0 = time.time()
        N.check(lib.lia_host_layer_forward(ctypes.byref(m.desc), ctypes.byref(w), ctypes.c_void_p(hx), ctypes.c_void_p(hy),
                                           ctypes.c_void_p(kv.k), ctypes.c_void_p(kv.v), kv.smax, kv.batch, B, T, pos0, 0, threads),
                "lia_host_layer_forward")
        self.host_layer_time["ms"] += 1e3 * (time.time() - t0)
        self.host_layer_time["calls"] += 1
        N.check(lib.lia_blit(ctypes.c_void_p(y.data_ptr()), ctypes.c_void_p(hy), nbytes, ctypes.c_void_p(ctx.stream)), "lia_blit")
        return y, x

    def _host_layers(self, x, kv_state, n_gpu, B, T, pos0):
        """Policy 1 ("compute everything on CPU", lia/modeling_opt.py:1168, branches :1367-1377 / :1545-1555): layers
        [0, n_gpu) still run on the GPU (policy 3), the hidden state then moves to pinned host memory ONCE
        (:1262-1267) and every remaining layer runs on the host cores (lia_host_layer_forward) straight from the
        host copy of its weights -- nothing is streamed."""
        m, sh, ctx = self.model, self.model.shape, self.ctx
        lib = ctx.lib
        y = torch.empty_like(x)
        for idx in range(n_gpu):
            ctx.layer_forward(m.desc, 3, self._resident(idx), x, y, kv_state.kv[idx], B, T, pos0, 0)
            x, y = y, x
        ctx.synchronize()
        nbytes = B * T * sh.hidden * 2
        hx, hy = self._hidden_pair(nbytes)
        N.check(lib.lia_memcpy_d2h(ctypes.c_void_p(hx), ctypes.c_void_p(x.data_ptr()), nbytes), "lia_memcpy_d2h")
        from . import hostinfo
        threads = self._host_team(self.dp.world if self.dp else 1).threads
        host = list(range(n_gpu, sh.layers))
        import time
        t_host0 = time.time()
        raw_ptr = {}
        for idx in host:
            raw_ptr[idx] = m.layers[idx].raw_host_ptr()
            if raw_ptr[idx] is None:
                raise ValueError("policy 1 needs the raw bf16 host copy, but the streamed layers were pinned in the pack10 wire "
                                 "format only by an earlier call; reload the model or set LIA_STREAM_FORMAT=raw")
        if host and B * T <= 256:
            # a decode step: every host layer in ONE OpenMP region (lia_host_layers_forward); the pointer tables are rebuilt only
            # when a layer's host copy or the caches moved
            key = (tuple(raw_ptr[i] for i in host), kv_state.serial, kv_state.version,
                   tuple(kv_state.kv[i].k for i in host), tuple(kv_state.kv[i].v for i in host))
            tab = getattr(self, "_host_tables", None)
            if tab is None or tab[0] != key:
                n = len(host)
                wt = (ctypes.c_void_p * (16 * n))()
                for j, i in enumerate(host):
                    base = raw_ptr[i]
                    for t in range(16):
                        wt[16 * j + t] = base + m.offsets[t]
                kt = (ctypes.c_void_p * n)(*[kv_state.kv[i].k for i in host])
                vt = (ctypes.c_void_p * n)(*[kv_state.kv[i].v for i in host])
                tab = self._host_tables = (key, wt, kt, vt)
            kv0 = kv_state.kv[host[0]]
            where = lib.lia_host_layers_forward(ctypes.byref(m.desc), len(host), tab[1], ctypes.c_void_p(hx), ctypes.c_void_p(hy), tab[2], tab[3],
                                                kv0.smax, kv0.batch, B, T, pos0, 0, threads)
            N.check(min(where, 0), "lia_host_layers_forward")
            if where == 1:
                hx, hy = hy, hx
        else:
            for idx in host:
                w = ops.weight_ptr_array(raw_ptr[idx], m.offsets)
                kv = kv_state.kv[idx]
                N.check(lib.lia_host_layer_forward(ctypes.byref(m.desc), ctypes.byref(w), ctypes.c_void_p(hx), ctypes.c_void_p(hy),
                                                   ctypes.c_void_p(kv.k), ctypes.c_void_p(kv.v), kv.smax, kv.batch, B, T, pos0, 0, threads),
                        "lia_host_layer_forward")
                hx, hy = hy, hx
        self.host_layer_time["ms"] += 1e3 * (time.time() - t_host0)
        self.host_layer_time["calls"] += len(host)
        N.check(lib.lia_memcpy_h2d(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(hx), nbytes), "lia_memcpy_h2d")
        return x

    def stream_stats(self, reset=False):
        return self.pipe.stats(reset) if self.pipe else (0.0, 0.0)

    def decode_stats(self, reset=False, block=True):
        return self.pipe.decode_stats(reset, block) if self.pipe else {"launches": 0, "ms": 0.0, "bytes_in": 0.0, "bytes_out": 0.0}

    def close(self):
        if self.pipe:
            self.pipe.close()
            self.pipe = None
        if self.ctx:
            self._await_all_deliveries()
            self._release_hidden_pair()
            self.ctx.close()
            self.ctx = None
