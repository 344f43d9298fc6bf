"""The build's on-disk streaming format and the twin of the reference's dummy-weight generator.

Reference: a model reaches the hot path as an HF checkpoint directory -- written by llm/utils/opt-weight-gen.py:61-69
(`save_pretrained`, U[0,1) dummy weights for opt-66b / opt-175b) or downloaded -- loaded with `from_pretrained(torch_dtype=bf16)`
(llm/single_instance/run_generation.py:159-166), TPP-blocked by ipex.optimize, and un-blocked again on the GPU at every use.
Here a model is written ONCE as a directory of per-layer wire buffers, exactly the bytes the weight streamer moves:

    <dir>/lia_model.json     format tag, model shape, the 16 tensor offsets inside a layer buffer, one entry per layer
                             {file, wire (0 raw | 10 pack10), bytes, raw_bytes}, the head file
    <dir>/head.bin           embed_tokens | embed_positions | final_ln_w | final_ln_b, raw bf16
    <dir>/layer_NNN.bin      one packed layer: raw bf16 (lia_layer_pack_offsets layout) or its lossless pack10 encoding

and loaded without a second host copy: resident layers are read straight into HBM, streamed layers are `mmap`ed (read-only,
shared) and the mapping is registered with the driver (hipHostRegister, read-only), so the copy engine DMAs out of the page
cache in the wire format on disk.  Every mapping is checked against the container's memory limit first (hostinfo).
"""
import json
import os

import numpy as np
import torch

from . import _native as N
from .model import LAYER_TENSORS, LayerStore, LiaOPTModel, OPTShape, draw_head, draw_layer, resolve_shape

FORMAT = "lia-packed-v1"
MANIFEST = "lia_model.json"


def is_packed_dir(path):
    return os.path.isfile(os.path.join(path, MANIFEST))


def _shape_dict(shape):
    return {"name": shape.name, "hidden": shape.hidden, "heads": shape.heads, "ffn": shape.ffn, "layers": shape.layers,
            "vocab": shape.vocab, "max_pos": shape.max_pos, "ln_eps": shape.ln_eps}


def _write_layer(path, flat_u8_dev, wire, scratch):
    """flat_u8_dev: the raw layer on the device.  Returns (wire actually used, bytes written)."""
    nbytes = flat_u8_dev.numel()
    enc = scratch._encode_packed(wire, flat_u8_dev) if wire else None
    src, n = (enc[0], enc[1]) if enc else (flat_u8_dev, nbytes)
    host = np.empty(n, np.uint8)
    N.check(N.lib().lia_memcpy_d2h(host.ctypes.data, src.data_ptr(), n), "lia_memcpy_d2h")
    with open(path, "wb") as f:
        host.tofile(f)
    return (wire if enc else 0), n


def _write_head(dirpath, tok, pos, lnw, lnb):
    parts = [t.contiguous().view(torch.int16).cpu().numpy().view(np.uint16).reshape(-1) for t in (tok, pos, lnw, lnb)]
    with open(os.path.join(dirpath, "head.bin"), "wb") as f:
        for p in parts:
            p.tofile(f)
    return [int(p.size) for p in parts]


def _finish(dirpath, shape, offsets, layer_bytes, layers, head_sizes, extra):
    man = {"format": FORMAT, "family": "opt", "shape": _shape_dict(shape), "tensors": list(LAYER_TENSORS), "offsets": [int(o) for o in offsets],
           "layer_bytes": int(layer_bytes), "layers": layers, "head": {"file": "head.bin", "elements": head_sizes}}
    man.update(extra or {})
    with open(os.path.join(dirpath, MANIFEST), "w") as f:
        json.dump(man, f, indent=1)
    return man


def save_packed(model, dirpath, wire=10):
    """Write a LiaOPTModel (any tiers) as a packed directory."""
    os.makedirs(dirpath, exist_ok=True)
    scratch = LayerStore(model.desc, model.offsets, model.layer_bytes)
    layers = []
    for i, st in enumerate(model.layers):
        w, n = _write_layer(os.path.join(dirpath, f"layer_{i:03d}.bin"), st._raw_on_device(), wire, scratch)
        layers.append({"file": f"layer_{i:03d}.bin", "wire": w, "bytes": n, "raw_bytes": int(model.layer_bytes)})
    head = _write_head(dirpath, model.embed_tokens, model.embed_positions, model.final_ln_w, model.final_ln_b)
    return _finish(dirpath, model.shape, model.offsets, model.layer_bytes, layers, head, None)


def write_dummy_checkpoint(shape, dirpath, seed=0, init="uniform01", wire=10, progress=None):
    """Twin of llm/utils/opt-weight-gen.py (create_opt_model + save_model, :43-69): a model of the named shape with dummy
    weights -- torch.rand_like on every parameter there, the same U[0,1) recipe here but SEEDED -- written layer by layer (one
    layer in HBM at a time: opt-175b is 333 GB of bf16), directly in the streaming format.  init="normal" gives the HF
    _init_weights model of run_generation's random-init instead.  The values are those of
    LiaOPTModel.random_init(shape, seed, init): a run from the directory and a run from the generator agree bit for bit."""
    os.makedirs(dirpath, exist_ok=True)
    m = LiaOPTModel(shape)
    scratch = LayerStore(m.desc, m.offsets, m.layer_bytes)
    layers = []
    for li in range(shape.layers):
        flat = draw_layer(shape, m.offsets, m.layer_bytes, li, seed, init).view(torch.uint8)
        w, n = _write_layer(os.path.join(dirpath, f"layer_{li:03d}.bin"), flat, wire, scratch)
        layers.append({"file": f"layer_{li:03d}.bin", "wire": w, "bytes": n, "raw_bytes": int(m.layer_bytes)})
        if progress:
            progress(li, n)
    head = _write_head(dirpath, *draw_head(shape, seed, init))
    return _finish(dirpath, shape, m.offsets, m.layer_bytes, layers, head, {"dummy": {"seed": seed, "init": init}})


def load_packed(dirpath, n_gpu_layers=0):
    """-> LiaOPTModel: layers [0, n_gpu_layers) in HBM, the others mapped from their files (tier "mapped", DMA-able, in the
    wire format on disk).  A later place() with other flags re-tiers them like any other layer."""
    man = json.load(open(os.path.join(dirpath, MANIFEST)))
    if man.get("format") != FORMAT or man.get("family") != "opt":
        raise ValueError(f"{dirpath}: not a {FORMAT} OPT directory")
    sh = man["shape"]
    shape = OPTShape(sh["name"], sh["hidden"], sh["heads"], sh["ffn"], sh["layers"], vocab=sh["vocab"], max_pos=sh["max_pos"],
                     ln_eps=sh.get("ln_eps", 1e-5))
    model = LiaOPTModel(shape)
    if list(model.offsets) != man["offsets"] or model.layer_bytes != man["layer_bytes"]:
        raise ValueError(f"{dirpath}: layer layout of the file differs from this library's lia_layer_pack_offsets")
    H = shape.hidden
    head = np.fromfile(os.path.join(dirpath, man["head"]["file"]), dtype=np.uint16)
    sizes = man["head"]["elements"]
    if head.size != sum(sizes) or sizes != [shape.vocab * H, (shape.max_pos + 2) * H, H, H]:
        raise ValueError(f"{dirpath}: head.bin does not match the model shape")
    cuts = np.cumsum([0] + sizes)

    def dev(a, *shp):
        return torch.from_numpy(a.view(np.int16)).view(torch.bfloat16).reshape(*shp).cuda()

    model.embed_tokens, model.embed_positions = dev(head[cuts[0]:cuts[1]], shape.vocab, H), dev(head[cuts[1]:cuts[2]], shape.max_pos + 2, H)
    model.final_ln_w, model.final_ln_b = dev(head[cuts[2]:cuts[3]], H), dev(head[cuts[3]:cuts[4]], H)
    # the streamed layers stay registered (pinned) for the life of the model: judge the whole plan before the first mapping,
    # LayerStore.set_from_mapped_file guards every single one again (an opt-175b directory is ~220 GB of them)
    from . import hostinfo
    hostinfo.check_host_allocation(sum(int(e["bytes"]) for e in man["layers"][n_gpu_layers:]),
                                   f"{dirpath}: registering {len(man['layers']) - n_gpu_layers} streamed layers")
    bad = sorted({int(e["wire"]) for e in man["layers"]} - {0, 10})
    if bad:
        # (pack11 / pack12 were choices of --wire until r04; their decode kernels are gone)
        raise ValueError(f"{dirpath}: layers in wire format {bad} -- this build reads raw (0) and pack10 (10) only; rewrite the directory "
                         "(python -m lia_amd.packed_checkpoint --model ... --wire pack10)")
    for i, (st, ent) in enumerate(zip(model.layers, man["layers"])):
        path = os.path.join(dirpath, ent["file"])
        if os.path.getsize(path) != ent["bytes"]:
            raise ValueError(f"{path}: {os.path.getsize(path)} bytes on disk, manifest says {ent['bytes']}")
        if i < n_gpu_layers:
            st.set_from_file_to_device(path, 0, ent["bytes"], ent["wire"])       # plain read -> HBM: never mapped, never registered
        else:
            st.set_from_mapped_file(path, 0, ent["bytes"], ent["wire"])
    torch.cuda.synchronize()
    return model


def main(argv=None):
    """`python -m lia_amd.packed_checkpoint --model opt-175b --save_dir DIR` -- the reference generator's two flags
    (opt-weight-gen.py:78-80), plus the seed / recipe / wire format."""
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", type=str, default="opt-66b")
    ap.add_argument("--save_dir", type=str, required=True)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--init", default="uniform01", choices=["uniform01", "normal", "trained-like"])
    ap.add_argument("--wire", default="pack10", choices=["raw", "pack10"])
    a = ap.parse_args(argv)
    shape = resolve_shape(a.model)
    wire = {"raw": 0, "pack10": 10}[a.wire]
    total = [0]

    def progress(li, n):
        total[0] += n
        print(f"layer {li + 1}/{shape.layers}: {n / 2**20:.1f} MiB ({total[0] / 2**30:.2f} GiB so far)", flush=True)

    man = write_dummy_checkpoint(shape, a.save_dir, a.seed, a.init, wire, progress)
    raw = sum(e["raw_bytes"] for e in man["layers"])
    print(f"Model saved to {a.save_dir}: {total[0] / 2**30:.2f} GiB of layers for {raw / 2**30:.2f} GiB of bf16 "
          f"({16.0 * total[0] / raw:.2f} bits per value)")
    return man


if __name__ == "__main__":
    main()
