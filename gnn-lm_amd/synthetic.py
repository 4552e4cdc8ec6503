"""Seeded synthetic workloads with the real shapes: there is no network, so
WikiText-103 / checkpoints are replaced by random tables of the same layout and dtype.

Value distributions: codes uint8 uniform; centroids N(0, 0.5^2); OPQ matrix N(0,1)/sqrt(d) (decode
only needs a dense matrix); tgt features N(0,1) -> fp16; neighbour ids uniform over [0, N) with
0.1% forced -1 and one token per block with no neighbour at all; kNN similarities cosine-like
U(0.2, 0.9) sorted descending; vals Zipf(1.0); targets drawn from the retrieved neighbours' vals with
p = 0.3 (so recall is non-trivial) else Zipf.
"""
import math

import numpy as np
import torch

from .adaptive_softmax import AdaptiveSoftmax
from .engine import BlockBatch, GnnLmEngine
from .hgt import HGT, CodeStore


def zipf_tokens(rs, vocab, size):
    """Zipf(1.0) over [0, vocab): inverse-CDF sampling of p(r) ~ 1/(r+1)."""
    u = rs.random_sample(size)
    h = math.log(vocab + 1.0)
    return np.minimum((np.exp(u * h) - 1.0).astype(np.int64), vocab - 1)


def zipf_dev(n, vocab, gen, dev):
    """Zipf(1.0) tokens generated on the device (the 103 M-row label table of the full-size workloads)."""
    u = torch.rand(n, generator=gen, device=dev, dtype=torch.float64)
    return torch.clamp((torch.exp(u * math.log(vocab + 1.0)) - 1.0).to(torch.int64), max=vocab - 1)


def device_codes(n_local, M, dev, seed):
    """uint8 i.i.d. uniform code table [n_local, M] generated on the device in 4 Mi-row pieces (WikiText-103:
    103,227,021 x 128 B = 13.2 GB, too large to ship from the host): bench.py and the full-size parity tests
    build the store with this one function, so they index the same bytes."""
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    codes = torch.empty(n_local, M, dtype=torch.uint8, device=dev)
    step = 1 << 22
    for s in range(0, n_local, step):
        e = min(n_local, s + step)
        codes[s:e] = torch.randint(0, 256, (e - s, M), generator=gen, device=dev, dtype=torch.uint8)
    return codes


def synthetic_ivfpq_index(N, d, nlist, M, dev, nprobe=32, seed=7, skew=0.0, **kw):
    """Shape-true, content-free IVF-PQ index of the reference's kNN index family (OPQ64_1024,IVF4096,PQ64 over the
    103 M WikiText-103 keys: 6.6 GB of codes): random codes and centroids, a random rotation; lists of equal length, or
    (``skew`` > 0) log-normally distributed lengths with that sigma -- k-means lists of real keys are skewed (the 8.4 M-key
    clustered test index: 1 .. 32,659 keys, median 1,349).  For search-throughput measurements only (bench.py,
    tools/ivfpq_bench.py); a real index comes from run_index_build."""
    from .ivfpq import IVFPQIndex
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    R = torch.linalg.qr(torch.randn(d, d, generator=g, device=dev, dtype=torch.float32))[0].contiguous()
    coarse = (torch.randn(nlist, d, generator=g, device=dev) / d ** 0.5).contiguous()
    pq = (torch.randn(M, 256, d // M, generator=g, device=dev) * 0.05).contiguous()
    if skew > 0:
        w = torch.exp(skew * torch.randn(nlist, generator=g, device=dev, dtype=torch.float64))
        sizes = torch.floor(w / w.sum() * N).to(torch.int64)
        sizes[0] += N - int(sizes.sum().item())                              # the rounding remainder
        off = torch.zeros(nlist + 1, device=dev, dtype=torch.int64)
        off[1:] = torch.cumsum(sizes, 0)
    else:
        per = -(-N // nlist)
        off = torch.clamp(torch.arange(nlist + 1, device=dev, dtype=torch.int64) * per, max=N)
    return IVFPQIndex(R, coarse, pq, off, torch.arange(N, device=dev, dtype=torch.int64), device_codes(N, M, dev, seed),
                      nprobe=nprobe, cosine=True, **kw)


def make_codec(rs, M, dsub, d, opq=True):
    cen = (rs.randn(M, 256, dsub) * 0.5).astype(np.float32)
    dpq = M * dsub
    A = (rs.randn(dpq, d) / np.sqrt(dpq)).astype(np.float32) if opq else None
    b = (rs.randn(dpq) * 0.1).astype(np.float32) if opq else None
    return cen, A, b


def make_asm_weights(rs, vocab, d, cutoff, factor=4):
    cut = list(cutoff) + ([vocab] if vocab > cutoff[-1] else [])
    emb, proj, prev = [], [], 0
    for i, c in enumerate(cut):
        dim = int(d // (factor ** i))
        emb.append(torch.from_numpy((rs.randn(c - prev, dim) * dim ** -0.5).astype(np.float32)))
        proj.append(None if i == 0 else torch.from_numpy((rs.randn(d, dim) * d ** -0.5).astype(np.float32)))
        prev = c
    class_proj = torch.from_numpy((rs.randn(len(cut) - 1, d) * d ** -0.5).astype(np.float32))
    return {"cutoff": cut, "emb": emb, "proj": proj, "class_proj": class_proj}


def make_block(rs, n_store, vals, vocab, d, T, kg, k, n_blocks=1):
    n = n_blocks * T
    ids = rs.randint(0, n_store, size=(n, kg)).astype(np.int64)
    ids[rs.random_sample(ids.shape) < 0.001] = -1
    ids[np.arange(n_blocks) * T + min(3, T - 1)] = -1          # one token per block without neighbours
    tgt_feats = rs.randn(n, d).astype(np.float16)
    knn_ids = rs.randint(0, n_store, size=(n, k)).astype(np.int64)
    knn_ids[::7, -2:] = -1
    knn_sims = -np.sort(-rs.uniform(0.2, 0.9, size=(n, k)).astype(np.float32), axis=1)
    hit = rs.random_sample(n) < 0.3
    targets = np.where(hit, vals[np.maximum(knn_ids[:, 0], 0)].astype(np.int64), zipf_tokens(rs, vocab, n))
    return {"ids": ids, "tgt_feats": tgt_feats, "targets": targets, "knn_sims": knn_sims, "knn_ids": knn_ids,
            "n_blocks": n_blocks, "T": T}


def make_problem(n_store, d, n_heads, M, dsub, vocab, cutoff, T, kg, left, right, n_layers, k, seed=1234,
                 n_blocks=1, opq=True):
    """Host-side (numpy / CPU torch) description of a small problem: consumed by the HIP engine and,
    in tests / smoke, by the oracle."""
    rs = np.random.RandomState(seed)
    cen, A, b = make_codec(rs, M, dsub, d, opq)
    codes = rs.randint(0, 256, size=(n_store, M)).astype(np.uint8)
    vals = zipf_tokens(rs, vocab, n_store).astype(np.int32)
    torch.manual_seed(seed)
    hgt = HGT(in_dim=d, hidden_dim=d, out_dim=d, n_layers=n_layers, n_heads=n_heads)
    with torch.no_grad():                                      # move LN / pri / biases off their init values
        for nm, p in hgt.named_parameters():
            if "norms" in nm or "relation_pri" in nm or nm.endswith("bias"):
                p.add_(torch.randn_like(p) * 0.1)
    return {"n_store": n_store, "d": d, "n_heads": n_heads, "n_layers": n_layers, "left": left, "right": right,
            "vocab": vocab, "codes": codes, "vals": vals, "cen": cen, "A": A, "b": b,
            "sd": {k_: v.detach().clone() for k_, v in hgt.state_dict().items()},
            "asm": make_asm_weights(rs, vocab, d, cutoff),
            "block": make_block(rs, n_store, vals, vocab, d, T, kg, k, n_blocks)}


def build_engine(prob, dev):
    t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    store = CodeStore(codes=t(prob["codes"]), centroids=t(prob["cen"]), n_store=prob["n_store"], vals=t(prob["vals"]),
                      A=t(prob["A"]), b=t(prob["b"]))
    hgt = HGT(in_dim=prob["d"], hidden_dim=prob["d"], out_dim=prob["d"], n_layers=prob["n_layers"],
              n_heads=prob["n_heads"])
    hgt.load_state_dict(prob["sd"])
    w = prob["asm"]
    asm = AdaptiveSoftmax(w["cutoff"], w["emb"], w["proj"], w["class_proj"], dev)
    return GnnLmEngine(hgt, asm, store, prob["left"], prob["right"])


def to_batch(blk, dev):
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return BlockBatch(ids=t(blk["ids"]), tgt_feats=t(blk["tgt_feats"]), targets=t(blk["targets"]),
                      n_blocks=blk["n_blocks"], T=blk["T"], knn_sims=t(blk["knn_sims"]), knn_ids=t(blk["knn_ids"]))


def run_hip_block(prob, dev, lmbda, temperature):
    eng = build_engine(prob, dev)
    out = eng.score(to_batch(prob["block"], dev), lmbda, temperature)
    torch.cuda.synchronize()
    return {k_: v.cpu().numpy() for k_, v in out.items()}
