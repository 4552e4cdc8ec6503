#!/usr/bin/env python3
"""Benchmark of the GNN+kNN eval hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
        (N > 1: one process per GPU -- either launched by torch.distributed.run, which sets WORLD_SIZE / RANK, or plainly:
         the script then starts its own N ranks as child processes before it touches the GPU and relays rank 0's line)

A step = one pass of the hot path (gather -> HGT -> adaptive softmax -> kNN interpolation -> score
sum) over one batch of `--blocks` independent 256-token blocks of synthetic input with the real
WikiText-103 shapes (BASELINE.json configs[1]): 103,227,021-key PQ store (128-B codes, OPQ 1024x1024)
and label table resident in HBM, d=1024, 8 heads, k_g=128 graph neighbours with context 2+2, kNN
k=1024 (search results given), vocabulary 267,744 with the tied adaptive softmax.
Inputs are resident in HBM when the timed region starts.  N > 1: every rank scores its own blocks
(weak scaling); the store is range-sharded and rows are fetched with an RCCL all-to-all
(--store replicated skips the exchange).

Prints ONE JSON line (rank 0).  `roofline` describes the kernel with the largest share of the step,
timed with HIP events on its launch stream inside the timed region; `kernels` lists every kernel of
the step from a profiled warm-up step; `cpu_baseline` is the CPU oracle (the reference algorithm as
written, torch-CPU fp32) on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

PEAK = {"mfma_f32_tflops": 157.3, "mfma_bf16_tflops": 2500.0, "hbm_gbs": 8000.0}   # MI355X_MICROARCH.md chip table
SPLIT_PRODUCTS = {"f32": 1, "bf16x3": 3, "bf16x6": 6}            # bf16 MFMA products issued per f32 multiply-add
KERNEL_BOUND = {"gemm_nt_f32_kernel": "mfma", "causal_attn_kernel": "mfma", "ivfpq_scan8_kernel": "mfma", "ivfpq_sums_kernel": "mfma"}     # everything else on this path: hbm
# rocprof kernel names (profiles/pmc_traffic.json keys) behind a profile id of the library (csrc/common.h KernelId)
PMC_FAMILIES = {"gemm_nt_f32_kernel": ["gemm_nt_f32", "gemm_lse_astationary"],
                "ivfpq_scan8_kernel": ["ivfpq_scan8_kernel<false>"],      # (the "search:" rows of the file are tools/profile_search.sh's 8192-query launches)
                "ivfpq_sums_kernel": ["ivfpq_scan8_kernel<true>"]}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--blocks", type=int, default=None,
                    help="256-token blocks per step per GPU; default 128 for the 1-layer model, 16 for deeper ones (their workspace is 3.3 GB per block).  "
                         "(rounds 1-5: 32; with the search inside the step a batch of 128 blocks = 32768 queries fills "
                         "the search's 8-query groups and keeps a list's bytes in L2 across them: 548 k tokens/s against 513 k at 32, 535 k at 64, 547 k at 256)")
    ap.add_argument("--layers", type=int, default=1, help="HGT layers (configs[1]: 1; the shipped recipe: 3)")
    ap.add_argument("--n-store", type=int, default=103227021)
    ap.add_argument("--gcn-k", type=int, default=128)
    ap.add_argument("--k", type=int, default=1024)
    ap.add_argument("--tokens-per-sample", type=int, default=256)
    ap.add_argument("--lmbda", type=float, default=0.25)
    ap.add_argument("--temperature", type=float, default=0.01)
    ap.add_argument("--store", choices=["sharded", "replicated"], default="sharded")
    ap.add_argument("--pool", type=int, default=4, help="distinct input batches cycled through")
    ap.add_argument("--streams", type=int, default=1,
                    help="HIP streams per GPU: the step's blocks are split into this many independent sub-batches "
                         "enqueued on separate streams (HBM/L2-bound and MFMA-bound kernels of different sub-batches overlap)")
    ap.add_argument("--settle-s", type=float, default=0.6,
                    help="untimed back-to-back steps run before the timed region until this many seconds of GPU work have "
                         "passed AND two successive chunks agree within 2 %% (clock / power state settled)")
    ap.add_argument("--settle-max-s", type=float, default=6.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the oracle check of the first block before timing")
    ap.add_argument("--no-extras", dest="extras", action="store_false",
                    help="skip the recipe_L3 and driver_path sub-benchmarks (run after the timed region, N = 1 only)")
    ap.add_argument("--cpu-tokens", type=int, default=384)
    ap.add_argument("--l3-blocks", type=int, default=16, help="blocks per step of the 3-layer recipe sub-benchmark (recipe_L3)")
    ap.add_argument("--l3-steps", type=int, default=20)
    ap.add_argument("--l3-cache-gib", type=float, default=16.0, help="HBM budget of the cross-batch centre-state cache in recipe_L3")
    ap.add_argument("--search-check", type=int, default=32,
                    help="queries of the kNN-search sub-benchmark checked against the float64 oracle before it is timed (0: skip)")
    ap.add_argument("--small", action="store_true", help="tiny shapes (plumbing check only)")
    ap.add_argument("--graph", action="store_true",
                    help="capture each batch's step once in a HIP graph and replay it in the timed region (launch-bound "
                         "small batches; single GPU / replicated store only; the per-kernel roofline then comes from the "
                         "profiled warm-up step, the events cannot be read back from inside a graph)")
    ap.add_argument("--precision", choices=["f32", "bf16x3", "bf16x6"], default="f32",
                    help="GEMM arithmetic: f32 = native f32 MFMA (headline, the reference's precision); bf16x3 / bf16x6 = "
                         "opt-in split-bf16 emulation of the f32 product on the bf16 MFMA (max |dlogp| vs f32 is reported)")
    ap.add_argument("--shard-vals", action="store_true", help="also range-shard the label table (default: replicated)")
    ap.add_argument("--exchange", choices=["padded", "exact", "peer"], default="padded",
                    help="sharded store: fixed-capacity sync-free exchange (2 all-to-alls, no host round trip; dropped rows are "
                         "checked for after the run) or the exact variable-split one (3 all-to-alls, 2 host syncs per table)")
    ap.add_argument("--ids", choices=["uniform", "searched"], default="uniform",
                    help="graph-neighbour ids of the timed batches: i.i.d. uniform over the store (the headline: worst-case locality, no two "
                         "context groups coincide) or SEARCHED over a clustered synthetic corpus (searched_neighbour_ids: what the pipeline's own "
                         "kNN producer returns; equal context groups exist and are merged -- with a sharded store BEFORE the exchange)")
    ap.add_argument("--search", choices=["device", "given"], default="device",
                    help="device (default): the kNN search of every step's own queries runs on the device INSIDE the timed step (synthetic "
                         "OPQ64_1024,IVF4096,PQ64 index over --n-store keys, nprobe 32, k = --k; one replica per rank), as it runs inside the "
                         "reference's timer (fairseq_cli/eval_lm.py:214-219 -> knn_model.py:100); given: the batches carry precomputed search "
                         "results (`value_search_given` of the default run).  --small / --graph / --shard-vals runs use `given`")
    ap.add_argument("--lanes", type=int, default=2,
                    help="--search device: batches in flight, each on its own HIP stream (the host looks at a search's survivor counts when it "
                         "comes back to that batch; the latency-bound kernels of one batch's search run beside another batch's GEMMs). "
                         "1 = one batch at a time on one stream.  A sharded store keeps 1 (the exchange's collectives stay on one stream)")
    ap.add_argument("--force-exchange", action="store_true",
                    help="run the sharded-store exchange even with one rank (exercises the RCCL path on one GPU)")
    args = ap.parse_args()
    if args.blocks is None:
        args.blocks = 128 if args.layers == 1 else 16
    return args


def build(args, dev, rank, world):
    from gnnlm_amd.adaptive_softmax import AdaptiveSoftmax
    from gnnlm_amd.dist import Shard
    from gnnlm_amd.engine import GnnLmEngine
    from gnnlm_amd.hgt import HGT, CodeStore
    from gnnlm_amd.synthetic import device_codes, make_asm_weights, make_codec, zipf_dev
    if args.small:
        d, H, M, dsub, vocab, cutoff = 128, 8, 16, 8, 5000, [500, 2000]
    else:
        d, H, M, dsub, vocab, cutoff = 1024, 8, 128, 8, 267744, [20000, 60000]
    rs = np.random.RandomState(1234)
    cen, A, b = make_codec(rs, M, dsub, d, opq=True)
    sharded = (world > 1 or args.force_exchange) and args.store == "sharded"
    # halo layout: every shard also holds the 2 rows before / after its range, so that a context group (l = r = 2) is ONE
    # request answered by its centre's owner (dist.exchange_fetch_groups; used when the HGT needs every slot, L > 1)
    shard = Shard(args.n_store, world if sharded else 1, rank if sharded else 0, halo_left=2, halo_right=2)
    n_local = shard.n_local
    codes = device_codes(shard.store_rows, M, dev, 1234 + shard.row0)     # uint8 i.i.d. uniform
    if sharded and world > 1:
        # the rows near a shard boundary exist on two ranks: give them content that depends on the global row only
        near = torch.cat([torch.arange(max(0, b_ - 4), min(args.n_store, b_ + 4)) for b_ in range(shard.per, args.n_store, shard.per)])
        near = near[(near >= shard.store_row0) & (near < shard.store_row0 + shard.store_rows)].to(dev)
        if near.numel():
            codes[near - shard.store_row0] = (((near.reshape(-1, 1) * 2654435761 + torch.arange(M, device=dev) * 40503) >> 7) & 255).to(torch.uint8)
    # the label table (413 MB for WikiText-103) is replicated on every rank unless --shard-vals: only the
    # 13.2-GB code table needs the range sharding, and replicated labels save the k=1024-per-token exchange
    shard_vals = sharded and args.shard_vals
    vgen = torch.Generator(device=dev)
    vgen.manual_seed(4321 + (shard.row0 if shard_vals else 0))
    vals = zipf_dev(shard.store_rows if shard_vals else args.n_store, vocab, vgen, dev).to(torch.int32)
    t = lambda a: torch.from_numpy(a).to(dev)
    store = CodeStore(codes=codes, centroids=t(cen), n_store=args.n_store, row0=shard.store_row0, vals=vals, A=t(A), b=t(b))
    store.vals_row0 = shard.store_row0 if shard_vals else 0
    torch.manual_seed(1234)
    hgt = HGT(in_dim=d, hidden_dim=d, out_dim=d, n_layers=args.layers, n_heads=H)
    # the headline cycles a small pool of synthetic batches: with the cross-batch centre-state cache on, every group would be a
    # hit from the second cycle on and `value` would be the all-hits floor, not the multi-layer step.  Off here; the cache has its
    # own, labelled lines (recipe_L3: cold pass / all hits)
    hgt.state_cache_gib = 0.0
    w = make_asm_weights(rs, vocab, d, cutoff)
    asm = AdaptiveSoftmax(w["cutoff"], w["emb"], w["proj"], w["class_proj"], dev)
    eng = GnnLmEngine(hgt, asm, store, 2, 2)
    from gnnlm_amd.ops import PRECISIONS
    hgt.gemm_precision = asm.gemm_precision = PRECISIONS[args.precision]
    cpu_model = {"sd": hgt.state_dict(), "asm": w, "cen": cen, "A": A, "b": b, "d": d, "H": H, "M": M, "vocab": vocab}
    return eng, shard, sharded, cpu_model, (d, vocab)


def make_batches(args, dev, rank, d, vocab):
    from gnnlm_amd.engine import BlockBatch
    from gnnlm_amd.synthetic import zipf_dev
    gen = torch.Generator(device=dev)
    gen.manual_seed(99 + rank)
    T, B, kg, k, N = args.tokens_per_sample, args.blocks // args.streams, args.gcn_k, args.k, args.n_store
    n = T * B
    out = []
    for _ in range(args.pool * args.streams):
        ids = torch.randint(0, N, (n, kg), generator=gen, device=dev, dtype=torch.int64)
        ids[torch.rand(n, kg, generator=gen, device=dev) < 0.001] = -1
        ids[torch.arange(B, device=dev) * T + 3] = -1
        feats = torch.randn(n, d, generator=gen, device=dev).half()
        knn_ids = torch.randint(0, N, (n, k), generator=gen, device=dev, dtype=torch.int64)
        knn_ids[::7, -2:] = -1
        sims = torch.sort(torch.rand(n, k, generator=gen, device=dev) * 0.7 + 0.2, dim=1, descending=True).values
        targets = zipf_dev(n, vocab, gen, dev)
        out.append(BlockBatch(ids=ids, tgt_feats=feats, targets=targets, n_blocks=B, T=T,
                              knn_sims=sims.contiguous(), knn_ids=knn_ids))
    if args.ids == "searched":
        kw = dict(n_keys=min(10000, N // 2), d_c=32, n_clusters=100) if args.small else {}
        rows, _ = searched_neighbour_ids(args, dev, n * len(out), T, seed=7 + rank, **kw)
        for i, b in enumerate(out):
            b.ids = rows[i * n:(i + 1) * n].contiguous()
    return out


def _drop_page_cache(path):
    """Best effort, no privileges needed: ask the kernel to forget the clean pages of one file."""
    try:
        fd = os.open(path, os.O_RDONLY)
        os.fsync(fd)
        os.posix_fadvise(fd, 0, 0, os.POSIX_FADV_DONTNEED)
        os.close(fd)
    except OSError:
        pass


def cpu_baseline(args, cpu_model):
    """The reference algorithm as written, timed stage by stage on the host (the oracle is
    test infrastructure, here it is the thing timed -- never the thing shipped):
      (1) np.memmap row gathers from files on local disk (neighbour ids, PQ codes, labels, fp16 features;
          cold = page cache dropped with posix_fadvise, warm = second read),
      (2) neighbour expansion: the reference's per-row Python loop (token_block_dataset.py:354-400) and its vectorised
          numpy equivalent,
      (3) PQ decode, (4) HGT x L un-elided in torch-CPU fp32, (5) adaptive softmax, (6) kNN prob + interpolation
    on every host core (`--cpu-tokens` tokens) and on ONE core (a 16-token sample; ~6 GFLOP per token and layer)."""
    import shutil
    import tempfile
    from gnnlm_amd.synthetic import make_block, zipf_tokens
    from oracle import adaptive_softmax as oas
    from oracle import graph as og
    from oracle import hgt as ohgt
    from oracle import knn as oknn
    from oracle import pq as opq
    rs = np.random.RandomState(7)
    n_host = min(args.n_store, 2_000_000)
    M, d, V = cpu_model["M"], cpu_model["d"], cpu_model["vocab"]
    Tc, kg, k, L = args.cpu_tokens, args.gcn_k, args.k, args.layers
    tmp = tempfile.mkdtemp(prefix="gnnlm_cpu_", dir=os.environ.get("TMPDIR", "/tmp"))
    try:
        vals_np = zipf_tokens(rs, V, n_host).astype(np.int32)
        blk = make_block(rs, n_host, vals_np, V, d, Tc, kg, k)
        files = {"codes": (rs.randint(0, 256, size=(n_host, M)).astype(np.uint8), np.uint8, (n_host, M)),
                 "vals": (vals_np, np.int32, (n_host,)),
                 "nbrs": (blk["ids"], np.int64, (Tc, kg)),
                 "feats": (blk["tgt_feats"], np.float16, (Tc, d))}
        for nm, (arr, _, _) in files.items():
            arr.tofile(os.path.join(tmp, nm))
        mm = {nm: np.memmap(os.path.join(tmp, nm), mode="r", dtype=dt, shape=sh) for nm, (_, dt, sh) in files.items()}
        del files
        sd = {k_: v.float() for k_, v in cpu_model["sd"].items()}
        asm = cpu_model["asm"]
        all_cores = torch.get_num_threads()

        def run(n_tok, cores, cold, L=L, sd=sd):
            torch.set_num_threads(cores)
            ms = {}
            clock = lambda: time.perf_counter()
            if cold:
                for nm in mm:
                    _drop_page_cache(os.path.join(tmp, nm))
            t0 = clock()
            nb = np.array(mm["nbrs"][:n_tok])
            feats = np.array(mm["feats"][:n_tok]).astype(np.float32)                      # token_block_dataset.py:327-329
            rows, valid = og.slot_layout(nb, n_host, 2, 2)
            flat = rows[valid]
            ncodes = np.array(mm["codes"][flat])                                          # :370-371,391-393 (row gathers)
            _ = np.array(mm["vals"][flat])                                                # neighbor_tokens[offset] (:410)
            ms["gather_memmap_" + ("cold" if cold else "warm")] = (clock() - t0) * 1e3
            t0 = clock()
            graph = og.build_graph(nb, np.zeros(n_tok, np.int64), n_host, 2, 2)            # per-row loop, as written
            ms["expand_reference_loop"] = (clock() - t0) * 1e3
            t0 = clock()
            og.slot_layout(nb, n_host, 2, 2)
            ms["expand_vectorised"] = (clock() - t0) * 1e3
            t0 = clock()
            ntgt = opq.pq_decode(ncodes, cpu_model["cen"], cpu_model["A"], cpu_model["b"])
            ms["pq_decode"] = (clock() - t0) * 1e3
            t0 = clock()
            h = ohgt.hgt_forward(sd, L, cpu_model["H"], {"tgt": torch.from_numpy(feats), "ntgt": torch.from_numpy(ntgt)}, graph)
            ms["hgt"] = (clock() - t0) * 1e3
            t0 = clock()
            tgt = torch.as_tensor(blk["targets"][:n_tok]).long()
            lm = oas.target_log_prob(h["tgt"], tgt, asm).float()
            ms["adaptive_softmax"] = (clock() - t0) * 1e3
            t0 = clock()
            pk, _ = oknn.knn_target_prob(blk["knn_sims"][:n_tok], blk["knn_ids"][:n_tok], mm["vals"], tgt, args.temperature)
            oknn.combine_knn_and_vocab_probs(pk, lm, args.lmbda)
            ms["knn_interp"] = (clock() - t0) * 1e3
            total = sum(v for k_, v in ms.items() if k_ != "expand_vectorised")           # the path as written
            return {"tokens": n_tok, "cores": cores, "tokens_per_s": n_tok / (total / 1e3),
                    "stage_ms": {k_: round(v, 2) for k_, v in ms.items()}, "seconds": round(total / 1e3, 2)}

        run(min(Tc, 8), all_cores, cold=False)                # untimed: first-call costs of the torch-CPU operators
        full = run(Tc, all_cores, cold=True)
        warm = run(min(Tc, 64), all_cores, cold=False)
        one = run(min(Tc, 16), 1, cold=False)
        # the shipped 3-layer recipe (hgt_lm_wiki103_reproduce.sh:56) on the same cores: the CPU pair of `recipe_L3`
        sd3 = {k_: v.float() for k_, v in ohgt.init_hgt_weights(3, d, cpu_model["H"], seed=7).items()}
        l3 = run(min(Tc, 64), all_cores, cold=False, L=3, sd=sd3) if L == 1 else None
        torch.set_num_threads(all_cores)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return {"value": round(full["tokens_per_s"], 2), "unit": "tokens/s", "cores": all_cores, "kind": "port",
            "sample": f"{Tc} tokens of one block, k_g={kg}, l=r=2, L={L}, kNN k={k}, d={d}, {n_host}-row tables as np.memmap "
                      f"files on local disk (cold page cache), reference-style expansion loop, torch-CPU fp32, {full['seconds']} s",
            "all_cores": full, "all_cores_warm": warm, "one_core": one, "recipe_L3_all_cores": l3}


def verify_block(eng, batch, args, cpu_model, n_tok):
    """Parity gate of the bench itself: the first `n_tok` tokens of the first pooled block (causal attention makes a
    prefix independent of the rest) through the oracle -- the reference algorithm as written, torch-CPU float32 --
    against what the timed engine returns for them.  The rows the block touches are pulled from the device store."""
    from oracle import graph as og
    from oracle import pipeline
    from oracle.hostrows import HostRows
    st = eng.store
    out = eng.score(batch, args.lmbda, args.temperature)
    got = out["logp"][:n_tok].double().cpu().numpy()
    nb = batch.ids[:n_tok].cpu().numpy()
    rows, valid = og.slot_layout(nb, st.n_store, eng.left, eng.right)
    model = {"sd": {k_: v.detach().float().cpu() for k_, v in cpu_model["sd"].items()}, "n_layers": args.layers,
             "n_heads": cpu_model["H"], "centroids": cpu_model["cen"], "A": cpu_model["A"], "b": cpu_model["b"],
             "codes": HostRows(st.codes, rows[valid]), "vals": st.vals.cpu().numpy(), "n_store": st.n_store,
             "left": eng.left, "right": eng.right, "asm": cpu_model["asm"]}
    one = {"neighbor_idxs": nb, "tgt_feats": batch.tgt_feats[:n_tok].cpu().numpy(), "targets": batch.targets[:n_tok].cpu().numpy(),
           "knn_sims": batch.knn_sims[:n_tok].cpu().numpy(), "knn_ids": batch.knn_ids[:n_tok].cpu().numpy()}
    t0 = time.perf_counter()
    ref = pipeline.eval_block(one, model, args.lmbda, args.temperature)["logp"].double().numpy()
    err = float(np.abs(got - ref).max())
    res = {"tokens": n_tok, "max_abs_dlogp_vs_oracle": err, "score_sum_hip": float(got.sum()), "score_sum_oracle": float(ref.sum()),
           "tolerance": 1e-4, "oracle_seconds": round(time.perf_counter() - t0, 1)}
    assert err < 1e-4, f"bench parity gate failed: {res}"
    return res


def searched_neighbour_ids(args, dev, n_tokens, T, n_keys=1_000_000, d_c=256, n_clusters=4000, noise=0.9, stay=0.7, seed=7):
    """Neighbour ids as the pipeline's own producer makes them: an exact cosine kNN search (the `ExactIndex` behind
    `python -m gnnlm_amd.find_knn`, knn/find_knn.py:55-70) of a token stream over a CLUSTERED synthetic corpus -- `n_keys` keys
    around `n_clusters` centres; a block's tokens walk through the clusters (a token stays in its predecessor's cluster with
    probability `stay`: topical text), each token = its centre + noise.  Returns ids [n_tokens, k_g] as STORE rows (key i lives
    in row i * (n_store // n_keys), so context rows are distinct rows) and a description.  The overlap of the lists is a
    property of the search result, not of this function."""
    from gnnlm_amd.knn_model import ExactIndex
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    centres = torch.randn(n_clusters, d_c, generator=g, device=dev)
    which = torch.randint(0, n_clusters, (n_keys,), generator=g, device=dev)
    keys = (centres[which] + noise * torch.randn(n_keys, d_c, generator=g, device=dev)).to(torch.float16)
    index = ExactIndex(keys, metric="ip", cosine=True, device=dev)
    # the token stream: cluster of token t = cluster of t - 1 with probability `stay` (never across a block boundary)
    fresh = torch.randint(0, n_clusters, (n_tokens,), generator=g, device=dev)
    keep = torch.rand(n_tokens, generator=g, device=dev) < stay
    keep[::T] = False
    seg = torch.cumsum((~keep).to(torch.int64), 0) - 1                     # tokens of one stay-run share the run's first draw
    cl = fresh[(~keep).nonzero().reshape(-1)][seg]
    q = centres[cl] + noise * torch.randn(n_tokens, d_c, generator=g, device=dev)
    t0 = time.perf_counter()
    _, ids = index.search_device(q, args.gcn_k)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    stride = max(1, args.n_store // n_keys)
    rows = torch.where(ids >= 0, ids * stride, ids)
    desc = {"producer": "ExactIndex.search_device (what gnnlm_amd.find_knn runs), cosine", "corpus_keys": n_keys, "corpus_dim": d_c,
            "clusters": n_clusters, "noise": noise, "p_stay_in_cluster": stay, "search_seconds": round(dt, 2),
            "row_of_key": f"key i -> store row {stride} i"}
    del index, keys
    return rows.contiguous(), desc


def recipe_l3(args, eng1, batches, dev):
    """The shipped recipe (`--graph_layer 3`, hgt_lm_wiki103_reproduce.sh:56: the model the published 14.8 ppl belongs to) on
    the same store, as a measured configuration of its own: `--l3-blocks` blocks per step (blocks are independent), >= 20
    timed steps after a settle phase, per-kernel list, roofline of the dominant kernel with its PMC traffic, executed and
    un-elided FLOP per token.  Neighbour ids: (1) i.i.d. uniform over the store -- no two context groups coincide: the worst
    case, nothing to merge or cache; (2) SEARCHED neighbours over a clustered corpus (`searched_neighbour_ids`) -- equal groups
    of a batch are computed once and the centre states are cached across batches (exact: token_block_dataset.py:355)."""
    from gnnlm_amd import _lib, ops
    from gnnlm_amd.engine import BlockBatch, GnnLmEngine
    from gnnlm_amd.hgt import HGT
    torch.manual_seed(4321)
    d, H = eng1.hgt.hidden_dim, eng1.hgt.n_heads
    hgt = HGT(in_dim=d, hidden_dim=d, out_dim=d, n_layers=3, n_heads=H)
    hgt.gemm_precision = eng1.hgt.gemm_precision
    eng = GnnLmEngine(hgt, eng1.asm, eng1.store, eng1.left, eng1.right)
    nb, T = args.l3_blocks, args.tokens_per_sample
    n = nb * T
    pool_feats = torch.cat([b_.tgt_feats for b_ in batches])
    pool_tg, pool_sims, pool_kids = (torch.cat([getattr(b_, a) for b_ in batches]) for a in ("targets", "knn_sims", "knn_ids"))
    pool_ids = torch.cat([b_.ids for b_ in batches])
    assert pool_feats.shape[0] >= n, "--l3-blocks larger than the input pool (--pool x --blocks)"
    n_pool = pool_feats.shape[0] // n                                     # distinct L3 batches in the pool

    def batch_at(i, ids):
        sl = slice((i % n_pool) * n, (i % n_pool + 1) * n)
        return BlockBatch(ids=ids[sl] if ids.shape[0] >= (i % n_pool + 1) * n else ids[:n], tgt_feats=pool_feats[sl], targets=pool_tg[sl],
                          n_blocks=nb, T=T, knn_sims=pool_sims[sl], knn_ids=pool_kids[sl])

    acc = torch.zeros(1, device=dev, dtype=torch.float64)
    step = lambda b_: ops.masked_sum_f64(eng.score(b_, args.lmbda, args.temperature)["logp"], None, acc)
    steps = max(20, args.l3_steps)

    def timed(bs, cache_gib, settle_s=0.6, steps=steps):
        """`steps` timed steps over the batch list `bs` (cycled), after one profiled step and a settle phase."""
        hgt.state_cache_gib, hgt.state_cache = cache_gib, None
        step(bs[0])
        torch.cuda.synchronize()
        _lib.profile_begin()
        step(bs[1 % len(bs)])
        torch.cuda.synchronize()
        kern = _lib.profile_end()
        t_s, i = 0.0, 2
        while t_s < settle_s:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            step(bs[i % len(bs)])
            torch.cuda.synchronize()
            t_s += time.perf_counter() - t0
            i += 1
        dominant = max(kern, key=lambda k_: kern[k_]["total_ms"])
        names = [_lib.lib().gnnlm_kernel_name(j).decode() for j in range(12)]
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        groups = []
        torch.cuda.synchronize()
        _lib.profile_begin(1 << names.index(dominant))
        t0 = time.perf_counter()
        marks[0].record()
        for j in range(steps):
            step(bs[(i + j) % len(bs)])
            groups.append(hgt._last_groups)                               # (n, device counter): read after the loop, no sync inside it
            marks[j + 1].record()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        groups = [int(g_[1][0].item()) if g_ is not None else n * args.gcn_k for g_ in groups]
        prof = _lib.profile_end()[dominant]
        per = sorted(marks[j].elapsed_time(marks[j + 1]) for j in range(steps))
        r = roofline_entry(dominant, prof, args.precision)
        r["traffic"], r["traffic_source"] = pmc_traffic("L3:" + dominant)
        r["launches_per_step"] = prof["launches"] / steps
        flop_exec = sum(v["flops"] for v in kern.values()) / n
        return {"tokens_per_s": round(steps * n / dt, 1), "ms_per_step": round(dt / steps * 1e3, 3), "ms_per_step_median": round(per[len(per) // 2], 3),
                "ms_per_step_min": round(per[0], 3), "ms_per_step_max": round(per[-1], 3), "steps": steps,
                "context_groups_per_step": n * args.gcn_k, "groups_computed_per_step_mean": round(sum(groups) / len(groups), 1),
                "roofline": r, "kernels": [roofline_entry(k_, v, args.precision) for k_, v in sorted(kern.items(), key=lambda kv: -kv[1]["total_ms"])],
                "dominant_share_of_step": round(kern[dominant]["total_ms"] / sum(v["total_ms"] for v in kern.values()), 3),
                "flop_per_token_executed": round(flop_exec), "executed_TFLOPs_whole_step": round(flop_exec * steps * n / dt / 1e12, 2)}

    out = {"hgt_layers": 3, "blocks_per_step": nb, "tokens_per_step": n,
           "flop_per_token_unelided_survey_8d": round(unelided_flop_per_token(3, d=d, H=H, kg=args.gcn_k, T=T)),
           "flop_note": "`flop_per_token_executed` = what the kernels of a step compute (dead work elided, star edges absorbed: DESIGN.md 2); "
                        "the un-elided figure is the reference's forward as written (SURVEY.md 8d), the same result"}
    uni = [batch_at(i, pool_ids) for i in range(n_pool)]
    out["uniform_ids"] = dict(timed(uni, 0.0), neighbour_ids="i.i.d. uniform over the store (no equal context groups: the worst case; state cache off -- it could only miss)")
    out["uniform_ids"]["unelided_equivalent_TFLOPs"] = round(out["flop_per_token_unelided_survey_8d"] * out["uniform_ids"]["tokens_per_s"] / 1e12, 1)
    if args.precision == "f32":
        # the same worst case under the opt-in split-bf16 emulation of the f32 product (`--precision bf16x3`: three bf16 MFMA products per
        # multiply-add, f32 accumulate): NOT the headline's arithmetic -- reported with its measured distance from the f32 path
        lp32 = eng.score(uni[0], args.lmbda, args.temperature)["logp"].clone()
        saved = hgt.gemm_precision, eng.asm.gemm_precision
        hgt.gemm_precision = eng.asm.gemm_precision = ops.PRECISIONS["bf16x3"]
        try:
            lp3 = eng.score(uni[0], args.lmbda, args.temperature)["logp"]
            dl = float((lp3 - lp32).abs().max().item())
            t3 = timed(uni, 0.0, settle_s=0.3)
            out["uniform_ids_bf16x3_opt_in"] = {"tokens_per_s": t3["tokens_per_s"], "ms_per_step": t3["ms_per_step"], "ms_per_step_median": t3["ms_per_step_median"],
                                                "steps": t3["steps"], "max_abs_dlogp_vs_f32": dl,
                                                "what": "`--precision bf16x3` (split-bf16 emulation of the f32 product on the bf16 matrix cores, f32 accumulate); opt-in, not the headline's arithmetic"}
        finally:
            hgt.gemm_precision, eng.asm.gemm_precision = saved
    # searched neighbours: one id table for the whole pool
    n_srch = n_pool * n
    ids_s, desc = searched_neighbour_ids(args, dev, n_srch, T)
    srch = [batch_at(i, ids_s) for i in range(n_pool)]
    flat = ids_s.reshape(n_pool, -1)
    per_batch = [int(torch.unique(flat[i][flat[i] >= 0]).numel()) for i in range(n_pool)]
    merged = timed(srch, 0.0)                                             # within-batch merge only

    def cold_passes(bs):
        """Cross-batch cache: COLD passes over the batch list (what an eval run over new text does), each from an empty cache."""
        hgt.state_cache_gib, hgt.state_cache = float(args.l3_cache_gib), None
        colds_ = []
        for rep in range(3):                                              # pass 0 allocates (cache, workspace); passes 1, 2 are timed
            if hgt.state_cache is not None:
                hgt.state_cache.clear()
                hgt.state_cache.reset_stats()
            torch.cuda.synchronize()
            t0_ = time.perf_counter()
            for b_ in bs:
                step(b_)
            torch.cuda.synchronize()
            colds_.append(time.perf_counter() - t0_)
        return colds_, (dict(hgt.state_cache.stats) if hgt.state_cache is not None else None)
    colds, st = cold_passes(srch)
    cold = min(colds[1:])
    # and the steady state of a long run over text whose rows the cache already holds (every group a hit)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b_ in srch:
        step(b_)
    torch.cuda.synchronize()
    warm = time.perf_counter() - t0
    out["searched_neighbours"] = {
        "corpus": desc, "batches_in_pool": n_pool, "distinct_context_groups_per_batch": per_batch,
        "distinct_context_groups_whole_pool": int(torch.unique(flat[flat >= 0]).numel()), "context_groups_per_batch": n * args.gcn_k,
        "within_batch_merge": {k_: merged[k_] for k_ in ("tokens_per_s", "ms_per_step", "ms_per_step_median", "groups_computed_per_step_mean", "dominant_share_of_step")},
        "with_state_cache_cold_pass": {"tokens": n_srch, "tokens_per_s": round(n_srch / cold, 1), "seconds": round(cold, 3), "cache_gib": args.l3_cache_gib,
                                       "seconds_runs": [round(c_, 3) for c_ in colds[1:]],
                                       "groups_computed": None if st is None else st["computed"], "groups_looked_up": None if st is None else st["groups"]},
        "with_state_cache_all_hits": {"tokens": n_srch, "tokens_per_s": round(n_srch / warm, 1), "seconds": round(warm, 3),
                                      "what": "the same pool again with every group's states cached: the floor of the cached path (tgt side + cache reads only)"},
        "note": "cold pass = every batch of the pool once from an empty cache (an eval run over new text); a group's centre states are "
                "computed the first time its row is retrieved and read from HBM afterwards (exact)"}
    # ---- sensitivity: the merge / cache factors are a property of how much the neighbour lists of nearby tokens overlap, i.e. of the
    # corpus knob `p_stay_in_cluster` (how long a block dwells on one topic).  Three points instead of one; 0.7 is the line above
    sens, l3_ids = {}, {"p_stay_0.7": ids_s}
    for stay in (0.3, 0.9):
        ids_x, _ = searched_neighbour_ids(args, dev, n_srch, T, stay=stay)
        l3_ids[f"p_stay_{stay}"] = ids_x
        bs = [batch_at(i, ids_x) for i in range(n_pool)]
        fx = ids_x.reshape(n_pool, -1)
        m_x = timed(bs, 0.0, settle_s=0.3, steps=10)
        c_x, st_x = cold_passes(bs)
        sens[f"p_stay_{stay}"] = {"distinct_context_groups_per_batch_mean": round(sum(int(torch.unique(fx[i][fx[i] >= 0]).numel()) for i in range(n_pool)) / n_pool),
                                  "distinct_context_groups_whole_pool": int(torch.unique(fx[fx >= 0]).numel()),
                                  "within_batch_merge_tokens_per_s": m_x["tokens_per_s"], "with_state_cache_cold_pass_tokens_per_s": round(n_srch / min(c_x[1:]), 1)}
    sens["p_stay_0.7"] = {"distinct_context_groups_per_batch_mean": round(sum(per_batch) / len(per_batch)),
                          "distinct_context_groups_whole_pool": out["searched_neighbours"]["distinct_context_groups_whole_pool"],
                          "within_batch_merge_tokens_per_s": merged["tokens_per_s"], "with_state_cache_cold_pass_tokens_per_s": round(n_srch / cold, 1)}
    out["searched_neighbours"]["sensitivity"] = dict(sorted(sens.items()), context_groups_per_batch=n * args.gcn_k,
                                                     note="same corpus (4000 clusters, noise 0.9), same pool size; only the probability that a token stays "
                                                          "in its predecessor's cluster changes.  uniform_ids above is the limit of no overlap at all")
    # ---- the structure real kNN-LM retrieval shows on top of topical overlap: when token t retrieves datastore position p (a matching
    # context), token t + 1 tends to retrieve p + 1.  Then hardly two CENTRES coincide (p + 1 is a slot of t's group, not its centre): the
    # group-level merge and the centre-state cache find almost nothing, although neighbouring groups share 4 of their 5 ROWS
    # (token_block_dataset.py:378-400: a group is the +-2 window of its centre).  Layer 0's Q / K / V of a slot are a function of its code row
    # alone: layer 0's K / V are keyed by ROW (gnnlm_hgt_io_t.row_table, ABI 11; GNNLM_DEDUP_ROWS=0 switches it off) -- this line measures
    # the regime
    g2 = torch.Generator(device=dev)
    g2.manual_seed(97)
    kgc, p_follow = args.gcn_k, 0.8
    fresh = torch.randint(0, args.n_store - 1, (n_srch, kgc), generator=g2, device=dev, dtype=torch.int64)
    follow = torch.rand(n_srch, kgc, generator=g2, device=dev) < p_follow
    follow[::T] = False                                                   # a block's first token starts every run
    pos = torch.arange(n_srch, device=dev).view(-1, 1).expand(-1, kgc)
    start = torch.where(follow, torch.zeros_like(pos), pos).cummax(0).values
    ids_c = (torch.gather(fresh, 0, start) + (pos - start)).clamp_(max=args.n_store - 1)
    bs = [batch_at(i, ids_c) for i in range(n_pool)]
    m_c = timed(bs, 0.0, settle_s=0.3, steps=10)
    f0 = ids_c[:n].reshape(-1)
    slot_rows = (f0.view(-1, 1) + torch.arange(-eng.left, eng.right + 1, device=dev).view(1, -1)).reshape(-1)
    slot_rows = slot_rows[(slot_rows >= 0) & (slot_rows < args.n_store)]
    out["consecutive_retrieval"] = {
        "neighbour_ids": f"runs: neighbour j of token t + 1 = neighbour j of token t, plus one, with probability {p_follow} (else a fresh uniform id); runs restart at block starts",
        "context_groups_per_batch": n * kgc, "distinct_context_groups_first_batch": int(torch.unique(f0).numel()),
        "slot_rows_first_batch": int(slot_rows.numel()), "distinct_slot_rows_first_batch": int(torch.unique(slot_rows).numel()),
        "within_batch_merge_tokens_per_s": m_c["tokens_per_s"], "groups_computed_per_step_mean": m_c["groups_computed_per_step_mean"],
        "layer0_kv_keyed_by_row": bool(hgt.dedup_rows),
        "note": "group-level merging has nothing to merge here and the centre-state cache nothing to hit (16.0-16.1 k tokens/s either way with slot-keyed "
                "projections, GNNLM_DEDUP_ROWS=0); layer 0's K / V are computed once per distinct slot ROW (ABI 11): 10 of a group's 27 row-GEMMs shrink "
                "by distinct_slot_rows / slot_rows"}
    del ids_c, fresh, follow, pos, start, bs
    out["_l3_ids"] = l3_ids                                               # (popped by main: knn_search runs them through eval_lm with the search inside)
    hgt.state_cache = None
    torch.cuda.empty_cache()
    out["through_eval_lm"] = {"what": "eval_lm.main -> SequenceScorer.generate, the recipe's command line (--max-tokens 256: one-block batches, "
                                      "16 per launch for a multi-layer model), the reference's own timer (fairseq_cli/eval_lm.py:214-219) and the wall clock of main",
                              "uniform_ids": driver_path_l3(args, eng1, batches, dev),
                              "searched_neighbours": driver_path_l3(args, eng1, batches, dev, ids=ids_s, label="searched")}
    return out


def driver_path_l3(args, eng1, batches, dev, ids=None, label="uniform"):
    """`eval_lm.main` -> `SequenceScorer.generate` with the 3-layer model, the recipe's command line as it stands (one-block
    batches, `--batch-blocks` at its default: 16 per launch for a multi-layer model), over the whole input pool."""
    import contextlib
    import io
    from gnnlm_amd import eval_lm, ops
    from gnnlm_amd.hgt import HGT
    from gnnlm_amd.model import GnnLmModel
    torch.manual_seed(4321)
    d, H = eng1.hgt.hidden_dim, eng1.hgt.n_heads
    hgt = HGT(in_dim=d, hidden_dim=d, out_dim=d, n_layers=3, n_heads=H)
    st = eng1.store
    T = args.tokens_per_sample
    cat = lambda a: torch.cat([getattr(b_, a) for b_ in batches])
    feats, targets, nbrs, sims, kids = cat("tgt_feats"), cat("targets"), (ids if ids is not None else cat("ids")), cat("knn_sims"), cat("knn_ids")
    n = min(feats.shape[0], nbrs.shape[0])
    model = GnnLmModel(hgt, eng1.asm, None)
    model.make_store = lambda codes, n_store, device: st

    class Knn:
        pos = 0

        def interpolate(self, queries, targets_, lm_logp, t, lmbda, k=0):
            m = queries.shape[0]
            sl = slice(self.pos, self.pos + m)
            self.pos = (self.pos + m) % n
            return ops.knn_interp(lm_logp.contiguous(), sims[sl].contiguous(), kids[sl].contiguous(), targets_.long().contiguous(), t, lmbda,
                                  vals=st.vals, n_store=st.n_store)
    tabs = {"n_tok": n, "d": d, "vocab": None, "n_store": st.n_store, "feats": feats[:n], "targets": targets[:n].clamp(min=4),
            "nbrs": nbrs[:n], "codes": st.codes, "no_pad": True}
    a = eval_lm.get_parser().parse_args(
        ["-", "--path", "-", "--graph", "--use-precompute-feat", "--neighbor-context", "2", "--gcn-k", str(args.gcn_k),
         "--tokens-per-sample", str(T), "--max-tokens", str(T), "--knnlm", "--k", str(args.k), "--lmbda", str(args.lmbda),
         "--temperature", str(args.temperature), "--knn-keytype", "gcn_feat", "--softmax-batch", str(64 * T + 1), "--device", str(dev)])
    out = {}
    for name, gib in (("state_cache_off", 0.0), ("state_cache_on_cold", float(args.l3_cache_gib))):
        hgt.state_cache_gib, hgt.state_cache = gib, None
        with contextlib.redirect_stdout(io.StringIO()):
            a.knn_model = Knn()
            eval_lm.main(a, tables=tabs, model=model)                          # warm-up (allocations)
            if hgt.state_cache is not None:
                hgt.state_cache.clear()
            a.knn_model = Knn()
            r = eval_lm.main(a, tables=tabs, model=model)
        out[name] = {"tokens": r["tokens"], "tokens_per_s_generate_timer": round(r["tokens"] / r["seconds"], 1),
                     "tokens_per_s_wall": round(r["tokens"] / r["wall_seconds"], 1), "blocks_per_batch": a.batch_blocks}
        if label == "uniform":
            break                                                             # (i.i.d. ids: the cache could only miss)
    hgt.state_cache = None
    torch.cuda.empty_cache()
    return out


def driver_path(args, eng, batches, dev):
    """tokens/s of the DROP-IN path: `eval_lm.main` -> `SequenceScorer.generate` -> hypotheses -> score sum, the
    reference's own loop and timer (fairseq_cli/eval_lm.py:208-331: `seconds` = time inside generate, `wall` = the
    whole loop), one 256-token block per batch as in the recipe (--max-tokens 256), `--max-tokens` = the bench's
    batch, and the recipe's command as it stands (`--batch-blocks` at its default: the one-block batches, 32 per launch).  The tables are the bench's HBM-resident ones (the driver normally uploads them from the data directory)."""
    from gnnlm_amd import eval_lm
    from gnnlm_amd.model import GnnLmModel
    st = eng.store
    T, nblk = args.tokens_per_sample, min(args.blocks, 32)
    nb = nblk * T                                                             # tokens of one bench batch
    n = nb * len(batches)                                                     # the whole input pool: several batches per run

    class _Pool:                               # the pool's batches back to back (the driver slices blocks out of flat tables)
        tgt_feats = torch.cat([b.tgt_feats[:nb] for b in batches])
        targets = torch.cat([b.targets[:nb] for b in batches])
        ids = torch.cat([b.ids[:nb] for b in batches])
        knn_sims = torch.cat([b.knn_sims[:nb] for b in batches])
        knn_ids = torch.cat([b.knn_ids[:nb] for b in batches])
    b0 = _Pool
    model = GnnLmModel(eng.hgt, eng.asm, None)
    model.make_store = lambda codes, n_store, device: st                      # resident store, codec already folded

    class Replay:                              # search results given: faiss contract on device tensors
        def __init__(self):
            self.pos = 0

        def search_device(self, q, kk):
            m = q.shape[0]
            sl = slice(self.pos, self.pos + m)
            self.pos = (self.pos + m) % n
            return b0.knn_sims[sl, :kk], b0.knn_ids[sl, :kk]

    class Knn:                                 # KNNModel.interpolate contract over the resident label table
        def __init__(self):
            self.index = Replay()

        def interpolate(self, queries, targets, lm_logp, t, lmbda, k=0):
            from gnnlm_amd import ops
            sims, knns = self.index.search_device(queries, args.k)
            return ops.knn_interp(lm_logp.contiguous(), sims.contiguous(), knns.contiguous(), targets.long().contiguous(), t,
                                  lmbda, vals=st.vals, n_store=st.n_store)

    tabs = {"n_tok": n, "d": eng.hgt.hidden_dim, "vocab": None, "n_store": st.n_store, "feats": b0.tgt_feats[:n],
            "targets": b0.targets[:n].clamp(min=4), "nbrs": b0.ids[:n], "codes": st.codes, "no_pad": True}
    out = {}
    # one_block_per_batch: the recipe's literal batches, `--batch-blocks 0` -- successive batches in turn on 6 HIP streams (the driver's
    # default for single-block batches: a batch is a chain of ~45 small dependent launches, 0.56 ms on the device for 0.2 ms of work);
    # ..._one_stream: the same strictly one after the other; ..._graph_capture: forward + softmax replayed from HIP graphs on those streams
    for name, max_tokens, coalesce in (("one_block_per_batch", T, 0), ("one_block_per_batch_one_stream", T, 0), ("one_block_per_batch_graph_capture", T, 0),
                                       ("bench_batch", nb, 0), ("one_block_batches_coalesced", T, -1)):
        a = eval_lm.get_parser().parse_args(
            (["--graph-capture"] if name.endswith("graph_capture") else ["--streams", "1"] if name.endswith("one_stream") else []) +
            ["-", "--path", "-", "--graph", "--use-precompute-feat", "--neighbor-context", "2", "--gcn-k", str(args.gcn_k),
             "--tokens-per-sample", str(T), "--max-tokens", str(max_tokens), "--knnlm", "--k", str(args.k), "--lmbda",
             str(args.lmbda), "--temperature", str(args.temperature), "--knn-keytype", "gcn_feat", "--softmax-batch",
             str(nb + 1), "--device", str(dev), "--batch-blocks", str(coalesce)])
        a.knn_model = Knn()
        import contextlib
        import io
        with contextlib.redirect_stdout(io.StringIO()):
            eval_lm.main(a, tables=tabs, model=model)                          # warm-up
            runs = []
            for _ in range(5 if max_tokens == T and coalesce == 0 else 1):     # (a pass over the pool is 40-70 ms of one-block batches: median of 5)
                a.knn_model = Knn()
                runs.append(eval_lm.main(a, tables=tabs, model=model))
            r = sorted(runs, key=lambda r_: r_["wall_seconds"])[len(runs) // 2]
        out[name] = {"tokens": r["tokens"], "tokens_per_s_generate_timer": round(r["tokens"] / r["seconds"], 1),
                     "tokens_per_s_wall": round(r["tokens"] / r["wall_seconds"], 1), "blocks_per_batch": max(max_tokens // T, 32 if coalesce < 0 else coalesce)}
        if len(runs) > 1:
            out[name]["tokens_per_s_wall_runs"] = [round(r_["tokens"] / r_["wall_seconds"]) for r_ in runs]
            out[name]["streams"] = 1 if name.endswith("one_stream") else 6
    return out


def kernel_source_hash():
    """Identity of the kernel sources a PMC profile belongs to (csrc/ + the C header)."""
    import glob
    import hashlib
    h = hashlib.sha1()
    for f in sorted(glob.glob(os.path.join(ROOT, "gnn-lm_amd", "csrc", "*")) + [os.path.join(ROOT, "include", "gnnlm.h")]):
        if os.path.isfile(f) and not f.endswith("Makefile"):
            h.update(os.path.basename(f).encode())
            h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def search_check(idx, q, k, n_check):
    """Parity gate of the search bench: the first `n_check` queries through the device search against the float64 IVFADC
    oracle (oracle/ivfpq.py) over the SAME index arrays -- restricted to the lists those queries can probe (their 40 best
    lists each: the oracle selects its own 32 among them), which is what fits a host transfer at 103 M keys."""
    from oracle import ivfpq as oivf
    qs = q[:n_check].contiguous()
    t0 = time.perf_counter()
    v, i = idx.search_device(qs, k)
    cs = (qs @ idx.R.t()) @ idx.coarse.t()
    U = torch.unique(cs.topk(min(40, idx.nlist), dim=1).indices)
    lens = (idx.list_off[U + 1] - idx.list_off[U])
    off_sub = torch.zeros(U.numel() + 1, dtype=torch.int64, device=q.device)
    off_sub[1:] = torch.cumsum(lens, 0)
    rows = torch.repeat_interleave(idx.list_off[U] - off_sub[:-1], lens) + torch.arange(int(off_sub[-1]), device=q.device)
    arrs = [idx.R.cpu().numpy(), idx.coarse[U].cpu().numpy(), idx.pq.cpu().numpy(), off_sub.cpu().numpy(),
            idx.list_ids[rows].cpu().numpy(), idx.list_codes[rows].cpu().numpy()]
    t1 = time.perf_counter()
    v_ref, i_ref = oivf.search(qs.cpu().numpy(), *arrs, k=k, nprobe=idx.nprobe)
    t_oracle = time.perf_counter() - t1
    v, i = v.cpu().numpy(), i.cpu().numpy()
    same = float(np.mean([len(set(a) & set(b)) / k for a, b in zip(i, i_ref)]))
    dv = float(np.abs(v - v_ref).max())
    ok = same >= 0.998 and dv <= 2e-5
    out = {"queries": int(qs.shape[0]), "lists_on_host": int(U.numel()), "id_set_overlap": round(same, 5), "max_abs_dscore": dv,
           "tolerance": {"id_set_overlap": 0.998, "score": 2e-5}, "oracle_seconds": round(time.perf_counter() - t0, 1),
           "cpu_port_queries_per_s": round(qs.shape[0] / t_oracle, 2),
           "cpu_port_note": "oracle/ivfpq.py: numpy float64 restatement of IVFADC on one core over the same lists (NOT faiss, which the reference calls and this image lacks)",
           "ok": ok}
    if not ok:
        raise SystemExit(f"bench.py: the on-device kNN search disagrees with the oracle: {out}")
    return out


def knn_search(args, eng, batches, dev, step_ms, l3_ids=None, idx=None):
    """The kNN SEARCH the headline step leaves out (the reference runs it on the CPU with faiss inside its timer,
    knn_model.py:100 under fairseq_cli/eval_lm.py:214-219): the step's queries (the HGT features of one batch) through the
    on-device IVF-PQ search over a synthetic index of the reference's index shape (OPQ64_1024,IVF4096,PQ64, nprobe 32,
    k = --k) with as many keys as the store, the labels travelling with the results (no label gather in the step).
    Checked against the oracle first, then timed; per-kernel times by HIP events of one more call."""
    from dataclasses import replace
    from gnnlm_amd import _lib
    from gnnlm_amd.synthetic import synthetic_ivfpq_index
    built_here = idx is None
    if built_here:
        idx = synthetic_ivfpq_index(args.n_store, eng.hgt.hidden_dim, 4096, 64, dev, nprobe=32)
        idx.attach_vals(eng.store.vals)                                   # index key ids = store rows: payload = id << 24 | label
    q = eng.features(batches[0])
    q = q / q.norm(dim=1, keepdim=True)                                   # knn_model.py:181-184 (cosine index)
    check = search_check(idx, q, args.k, args.search_check) if (args.search_check > 0 and built_here) else None   # (the timed run's own gate otherwise)
    idx.search_device(q, args.k, return_vals=True)                        # same shapes as the timed call: no allocation inside it
    torch.cuda.synchronize()
    times = []
    for _ in range(3):                                                    # three timed calls; the report is their median
        t0 = time.perf_counter()
        sims, ids, kvals = idx.search_device(q, args.k, return_vals=True)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    dt = sorted(times)[1]
    st = {k_: (float(v_.item()) if torch.is_tensor(v_) else v_) for k_, v_ in idx.stats.items()}
    _lib.profile_begin()
    idx.search_device(q, args.k, return_vals=True)
    torch.cuda.synchronize()
    prof = _lib.profile_end()
    n = q.shape[0]
    # the step fed by the search: similarities, ids and LABELS come from the index (knn_vals given: no gather of vals[ids])
    b2 = replace(batches[0], knn_sims=sims.contiguous(), knn_ids=ids.contiguous(), knn_vals=kvals.contiguous())
    for _ in range(2):
        eng.score(b2, args.lmbda, args.temperature)
    torch.cuda.synchronize()
    _lib.profile_begin(1 << 7)                                            # K_KNN
    t1 = time.perf_counter()
    for _ in range(10):
        eng.score(b2, args.lmbda, args.temperature)
    torch.cuda.synchronize()
    step2 = (time.perf_counter() - t1) / 10
    knn_prof = _lib.profile_end().get("knn_interp_kernel", {"launches": 1, "total_ms": 0.0})
    # the search of a 16-block batch (the 3-layer recipe's step: recipe_L3.*.tokens_per_s_with_search)
    qh = q[: min(n, 4096)].contiguous()
    idx.search_device(qh, args.k, return_vals=True)
    torch.cuda.synchronize()
    th = []
    for _ in range(3):
        t0 = time.perf_counter()
        idx.search_device(qh, args.k, return_vals=True)
        torch.cuda.synchronize()
        th.append(time.perf_counter() - t0)
    ms_4096 = sorted(th)[1] * 1e3
    scan_name = "int8-MFMA filter + exact float32 re-score" if idx.tiles is not None else "float32"
    thr_lists, cap_now, thr_sample = idx.dense_probes, idx.cand_cap, getattr(idx, "threshold_sample", 1)
    # roofline of the search's dominant kernel, the int8-MFMA filter: one table byte per (query, key, sub-quantizer) goes
    # through LDS (ds_read_b64 of 8 queries' bytes) and through the matrix core (a byte of the MFMA's A operand = 32 int8 ops)
    filt = prof.get("ivfpq_scan8_kernel", {"total_ms": 0.0, "launches": 0})
    thr = prof.get("ivfpq_sums_kernel", {"total_ms": 0.0, "launches": 0})
    roof = filter_roofline(filt["total_ms"], filt["launches"], st["pairs"], pmc_traffic("ivfpq_scan8_kernel")[0]) if filt["total_ms"] > 0 else None
    # THROUGH THE DROP-IN DRIVER: eval_lm.main -> SequenceScorer.generate -> KNNModel.interpolate with the search inside the
    # reference's own timer (fairseq_cli/eval_lm.py:214-219), 32 one-block batches per launch as in `driver_path`
    drv = None
    if not args.small:
        import contextlib
        import io
        from gnnlm_amd import eval_lm, ops
        from gnnlm_amd.model import GnnLmModel
        T, nblk = args.tokens_per_sample, args.blocks
        nb = nblk * T
        pool = torch.cat([b_.tgt_feats[:nb] for b_ in batches]), torch.cat([b_.targets[:nb] for b_ in batches]), torch.cat([b_.ids[:nb] for b_ in batches])
        st_ = eng.store
        model = GnnLmModel(eng.hgt, eng.asm, None)
        model.make_store = lambda codes, n_store, device: st_

        class SearchKnn:                          # KNNModel.interpolate_begin / _finish over the device index (cosine: knn_model.py:181-184)
            def interpolate_begin(self, queries, k=0):
                qn = queries.float()
                qn = qn / (qn ** 2).sum(-1, keepdims=True).sqrt()
                return idx.search_begin(qn.contiguous(), args.k, return_vals=True)

            def interpolate_finish(self, h, targets, lm_logp, t, lmbda):
                sims_, ids_, vals_ = h.result()
                return ops.knn_interp(lm_logp.contiguous(), sims_.contiguous(), ids_.contiguous(), targets.long().contiguous(), t, lmbda,
                                      n_store=st_.n_store, knn_vals=vals_.contiguous())

            def interpolate(self, queries, targets, lm_logp, t, lmbda, k=0):
                return self.interpolate_finish(self.interpolate_begin(queries), targets, lm_logp, t, lmbda)
        n_tok = nb * len(batches)
        tabs = {"n_tok": n_tok, "d": eng.hgt.hidden_dim, "vocab": None, "n_store": st_.n_store, "feats": pool[0], "targets": pool[1].clamp(min=4),
                "nbrs": pool[2], "codes": st_.codes, "no_pad": True}
        a = eval_lm.get_parser().parse_args(
            ["-", "--path", "-", "--graph", "--use-precompute-feat", "--neighbor-context", "2", "--gcn-k", str(args.gcn_k),
             "--tokens-per-sample", str(T), "--max-tokens", str(T), "--knnlm", "--k", str(args.k), "--lmbda", str(args.lmbda),
             "--temperature", str(args.temperature), "--knn-keytype", "gcn_feat", "--softmax-batch", str(nb + 1), "--device", str(dev),
             "--batch-blocks", str(nblk)])
        a.knn_model = SearchKnn()
        with contextlib.redirect_stdout(io.StringIO()):
            eval_lm.main(a, tables=tabs, model=model)                          # warm-up
            r_ = eval_lm.main(a, tables=tabs, model=model)
        drv = {"tokens": r_["tokens"], "tokens_per_s_generate_timer": round(r_["tokens"] / r_["seconds"], 1),
               "tokens_per_s_wall": round(r_["tokens"] / r_["wall_seconds"], 1), "blocks_per_batch": nblk,
               "what": f"eval_lm.main with the recipe's --max-tokens 256 (one-block batches, --batch-blocks {nblk} per launch), kNN search on the device inside generate"}
        # the 3-layer recipe through the same driver with the search inside generate, on the searched-neighbour id sets of
        # recipe_L3's sensitivity points: within-batch merge only, and a cold pass of the cross-batch cache
        l3_drv = None
        if l3_ids:
            from gnnlm_amd.hgt import HGT
            torch.manual_seed(4321)
            hgt3 = HGT(in_dim=eng.hgt.hidden_dim, hidden_dim=eng.hgt.hidden_dim, out_dim=eng.hgt.hidden_dim, n_layers=3, n_heads=eng.hgt.n_heads)
            model3 = GnnLmModel(hgt3, eng.asm, None)
            model3.make_store = lambda codes, n_store, device: st_
            l3_drv = {}
            for label, ids_x in sorted(l3_ids.items()):
                n3 = min(ids_x.shape[0], pool[0].shape[0])
                tabs3 = dict(tabs, n_tok=n3, feats=pool[0][:n3], targets=pool[1][:n3].clamp(min=4), nbrs=ids_x[:n3])
                a.batch_blocks = -1                                             # (auto: 16 one-block batches per launch for a multi-layer model)
                a.softmax_batch = 64 * T + 1
                l3_drv[label] = {}
                for name, gib in (("state_cache_off", 0.0), ("state_cache_on_cold", float(args.l3_cache_gib))):
                    hgt3.state_cache_gib, hgt3.state_cache = gib, None
                    with contextlib.redirect_stdout(io.StringIO()):
                        a.knn_model = SearchKnn()
                        eval_lm.main(a, tables=tabs3, model=model3)                 # warm-up (allocations)
                        if hgt3.state_cache is not None:
                            hgt3.state_cache.clear()
                        r3 = eval_lm.main(a, tables=tabs3, model=model3)
                    l3_drv[label][name] = {"tokens": r3["tokens"], "tokens_per_s_generate_timer": round(r3["tokens"] / r3["seconds"], 1),
                                           "tokens_per_s_wall": round(r3["tokens"] / r3["wall_seconds"], 1)}
            hgt3.state_cache = None
            del hgt3, model3
            torch.cuda.empty_cache()
    # the same search over an index with SKEWED lists (log-normal lengths, sigma 0.7: ~30x between the shortest and the longest
    # of 4096 lists), as k-means lists of real keys are: the groups of a long list are long tasks
    skewed = None
    if not args.small:
        del idx, b2
        torch.cuda.empty_cache()
        idx2 = synthetic_ivfpq_index(args.n_store, eng.hgt.hidden_dim, 4096, 64, dev, nprobe=32, skew=0.7)
        idx2.attach_vals(eng.store.vals)
        idx2.search_device(q, args.k, return_vals=True)
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            idx2.search_device(q, args.k, return_vals=True)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        st2 = {k_: (float(v_.item()) if torch.is_tensor(v_) else v_) for k_, v_ in idx2.stats.items()}
        lens = (idx2.list_off[1:] - idx2.list_off[:-1])
        skewed = {"list_lengths_min_median_max": [int(lens.min().item()), int(lens.median().item()), int(lens.max().item())],
                  "ms_per_batch": round(sorted(ts)[1] * 1e3, 2), "ms_per_batch_runs": [round(t_ * 1e3, 2) for t_ in ts],
                  "pairs_per_query": round(st2["pairs"] / n), "survivors_per_query": round(st2["survivors"] / n),
                  "queries_searched_again": st2.get("requeried", 0), "queries_per_s": round(n / sorted(ts)[1], 1)}
        del idx2
        torch.cuda.empty_cache()
    return {"index": "synthetic OPQ64_1024,IVF4096,PQ64", "keys": args.n_store, "nprobe": 32, "k": args.k, "queries": n,
            "skewed_lists": skewed, "driver_with_search": drv,
            "recipe_L3_driver_with_search": (None if not l3_ids or args.small else dict(l3_drv, what="eval_lm.main, 3-layer model, the recipe's one-block batches (16 per "
                                             "launch), the IVF-PQ search of the step's own queries inside generate (the reference's timer spans it): tokens/s on the "
                                             "searched-neighbour id sets of recipe_L3.searched_neighbours.sensitivity")),
            "scan": scan_name, "threshold_lists": thr_lists, "threshold_sample": thr_sample, "cand_cap": cap_now, "ms_per_4096_queries": round(ms_4096, 3),
            "ms_per_batch": round(dt * 1e3, 2), "ms_per_batch_runs": [round(t_ * 1e3, 2) for t_ in times], "queries_per_s": round(n / dt, 1),
            "pairs_per_query": round(st["pairs"] / n), "survivors_per_query": round(st["survivors"] / n), "candidates_per_query": round(st["candidates"] / n), "queries_searched_again": st.get("requeried", 0),
            "kernels_ms": {k_: round(v_["total_ms"], 3) for k_, v_ in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"])},
            "roofline": roof, "parity": check,
            "step_with_search_labels_ms": round(step2 * 1e3, 4),
            "knn_interp_with_labels_us": round(knn_prof["total_ms"] * 1e3 / max(1, knn_prof["launches"]), 1),
            "tokens_per_s_step_plus_search": round(n / (dt + step2), 1)}


def filter_roofline(total_ms, launches, pairs, traffic=None):
    """Roofline line of the search's dominant kernel, the int8-MFMA filter: one table byte per (query, key, sub-quantizer) goes
    through LDS (ds_read_b64 of 8 queries' bytes) and through the matrix core (a byte of the MFMA's operand = 32 int8 ops).
    `pairs`: (query, key) pairs of all probed lists over the `launches` launches (device-side list lengths)."""
    lookups = pairs * 64.0
    sec = max(total_ms, 1e-9) / 1e3
    return {"kernel": "ivfpq_scan8_kernel (int8-MFMA filter over all probed lists)", "bound": "mfma",
            "achieved": round(lookups * 32 / sec / 1e12, 1), "peak": 5000.0, "unit": "TOP/s (int8)", "frac": round(lookups * 32 / sec / 1e12 / 5000.0, 4),
            "peak_note": "dense i8 MFMA = 2x bf16 (MI355X_MICROARCH.md); its measured 16x16x64 ceiling is 3944 TOP/s.  The sums run on "
                         "v_smfmac_i32_16x16x128_i8 (the constant selector is the 2:4-sparse operand): `achieved` counts the dense-equivalent "
                         "ops of the table bytes (32 per byte) and is priced against the DENSE peak",
            "frac_of_measured_ceiling": round(lookups * 32 / sec / 1e12 / 3944.0, 4),
            "lds_lookup_bytes_per_s_TB": round(lookups / sec / 1e12, 2), "lds_frac_of_150TBps": round(lookups / sec / 1e12 / 150.0, 4),
            "list_bytes_GBps": round(pairs / 8 * 64 / sec / 1e9, 1),
            "avg_us": round(total_ms * 1e3 / max(1, launches), 1), "launches": launches,
            "table_byte_lookups": lookups, "traffic": traffic,
            "traffic_note": "HBM bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE from the PMC passes, profiles/pmc_traffic.json): "
                            "the list bytes above are what the groups request; most of them hit in the XCD's L2"}


def roofline_entry(name, e, precision="f32"):
    """One kernel's roofline line from its event-timed profile entry (launches, total_ms, algorithmic flops / bytes)."""
    bound = KERNEL_BOUND.get(name, "hbm")
    sec = max(e["total_ms"], 1e-9) / 1e3
    if bound == "mfma" and precision != "f32":                    # priced in the bf16 MFMA flops actually issued
        a, p, unit = SPLIT_PRODUCTS[precision] * e["flops"] / sec / 1e12, PEAK["mfma_bf16_tflops"], "TFLOP/s"
    elif bound == "mfma":
        a, p, unit = e["flops"] / sec / 1e12, PEAK["mfma_f32_tflops"], "TFLOP/s"
    else:
        a, p, unit = e["bytes"] / sec / 1e9, PEAK["hbm_gbs"], "GB/s"
    out = {"kernel": name, "bound": bound, "achieved": round(a, 2), "peak": p, "unit": unit,
           "frac": round(a / p, 4), "launches": e["launches"], "avg_us": round(e["total_ms"] * 1e3 / max(1, e["launches"]), 2)}
    # both rates for the kernels that gather AND contract (star attention: code-row gather + f32-MFMA attention)
    if e["flops"] > 0 and e["bytes"] > 0:
        out["algorithmic_GBps"] = round(e["bytes"] / sec / 1e9, 1)
        out["algorithmic_TFLOPs"] = round(e["flops"] / sec / 1e12, 2)
        out["mfma_f32_frac"] = round(e["flops"] / sec / 1e12 / PEAK["mfma_f32_tflops"], 4)
    return out


def unelided_flop_per_token(L, d=1024, H=8, kg=128, l=2, r=2, T=256, M=128, dsub=8, head=20002, tails=((256, 40000), (64, 207744))):
    """the work of the reference's forward AS WRITTEN per evaluated token -- every ntgt node projected in every
    layer, the OPQ rotation of every decoded row, the dense head; what `flop_per_token_executed` is to be read against."""
    n, dk = kg * (1 + l + r), d // H
    layer = 8 * d * d * n + 8 * d * d + 8 * d * dk * n + 4 * d * dk + 4 * d * (kg + kg * (1 + 3 * (l + r)) + (T + 1) / 2)
    decode = 2 * (M * dsub) * d * n
    softmax = 2 * d * head                                        # + the target's tail band, a few MFLOP
    return L * layer + decode + softmax


def read_sclk(device_index, sysfs_only=False):
    """Best effort: the shader clock (MHz) the device reports right now -- sysfs `pp_dpm_sclk` (the line marked `*`), else
    `rocm-smi --showclocks` (a subprocess: not from inside the timed region); None when neither is readable as this user.
    Read while kernels are queued (the host runs ahead of the device) it is the clock UNDER LOAD; after a sync it is the idle clock."""
    import glob
    import re
    import subprocess
    try:
        # the node's sysfs lists every GPU of the host; ours is the one at the PCI address the runtime reports
        pr = torch.cuda.get_device_properties(device_index)
        addr = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}."
        for f in sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk")):
            if addr in os.path.realpath(os.path.dirname(f)):
                m = re.search(r"(\d+)\s*M[Hh]z\s*\*", open(f).read())
                if m:
                    return int(m.group(1))
    except (OSError, AttributeError, RuntimeError):
        pass
    if sysfs_only:
        return None
    try:
        out = subprocess.run(["rocm-smi", "-d", str(device_index), "--showclocks"], capture_output=True, text=True, timeout=10).stdout
        m = re.search(r"sclk clock level:?\s*\S*:?\s*\(?(\d+)\s*M[Hh]z", out)
        return int(m.group(1)) if m else None
    except (OSError, subprocess.SubprocessError):
        return None


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/pmc_traffic.json, written by
    tools/pmc_summary.py from separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs of this same
    command; FETCH_SIZE doubled for gfx950 as MI355X_MICROARCH.md prescribes).  PMC counters cannot be read
    from inside the timed run, so this is the launch-weighted average over the kernel's instantiations of
    the profiled build -- and ONLY if that build is this one: the file carries the hash of the kernel sources
    it was measured on; a stale profile gives (None, reason) instead of a number that no longer applies."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            t = json.load(f)
    except OSError:
        return None, "no profiles/pmc_traffic.json"
    meta = t.pop("_meta", {})
    if meta.get("kernel_source_hash") != kernel_source_hash():
        return None, f"profiles/pmc_traffic.json ({meta.get('profile', 'unstamped')}) was measured on other kernel sources"
    # the instantiations a profile id covers (csrc/common.h KernelId -> the kernels launched under it)
    fams = PMC_FAMILIES.get(kernel) or [kernel[:-len("_kernel")] if kernel.endswith("_kernel") else kernel]
    rows = [v for k, v in t.items() if any(k.startswith(f_) for f_ in fams)]
    n = sum(v["launches"] for v in rows)
    if not n:
        return None, "kernel not in the profile"
    return round(sum(v["launches"] * v["hbm_bytes_per_launch"] for v in rows) / n), meta.get("profile")


def launch_ranks(args):
    """`python bench.py --gpus N` started plainly (no WORLD_SIZE in the environment): start the N ranks ourselves -- one process
    per GPU under torch.distributed.run, the launch the driver's own command line names -- relay their output (rank 0 prints the
    JSON line) and return the launcher's exit code.  Runs BEFORE this process touches the GPU, and the ranks are CHILD
    processes (never an exec of a process that initialised HIP; `torch.cuda.device_count()` may initialise the HIP runtime in this
    parent -- harmless for child processes, and the reason nothing here may ever become an exec)."""
    import socket
    import subprocess
    one_gpu_test = os.environ.get("GNNLM_BENCH_BACKEND", "nccl") != "nccl"     # tests: N ranks on device 0 over gloo
    have = torch.cuda.device_count()
    if have < args.gpus and not one_gpu_test:
        print(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) are visible (one rank per GPU over RCCL)", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL / mapped shards across processes need it
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


_T0 = time.perf_counter()


def stage(name):
    """Progress on stderr (rank 0's JSON line owns stdout): which leg of the run a crash or a time-out belongs to."""
    print(f"[bench.py {time.perf_counter() - _T0:7.1f} s] {name}", file=sys.stderr, flush=True)


def main():
    args = parse()
    if os.environ.get("GNNLM_BENCH_WATCHDOG"):                   # debugging aid: dump every thread's Python stack and exit after N seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["GNNLM_BENCH_WATCHDOG"]), exit=True)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    # GNNLM_BENCH_BACKEND=gloo GNNLM_BENCH_DEVICE=0 (tests only): several ranks on ONE GPU, collectives staged through the host --
    # the whole sharded code path (halo shards, fetch streams, reductions) with a real second rank on a one-GPU box
    backend = os.environ.get("GNNLM_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        os.environ["GNNLM_TEST_HOST_STAGED"] = "1"
    local_rank = int(os.environ.get("GNNLM_BENCH_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} in the environment (unset it, or launch {args.gpus} ranks)")
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product path has no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1 or args.force_exchange:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29577")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    def all_reduce(t, op=None):
        op = op or dist.ReduceOp.SUM
        if backend == "nccl":
            dist.all_reduce(t, op=op)
        else:
            h = t.cpu()
            dist.all_reduce(h, op=op)
            t.copy_(h)

    transport = dist.get_backend() if dist.is_initialized() else None   # what the collectives really run on: "nccl" (= RCCL) or the tests' "gloo"
    from gnnlm_amd import _lib, ops
    from gnnlm_amd.dist import PeerMappedFetcher, ShardedFetcher
    if rank == 0:
        stage("build: store, model")
    eng, shard, sharded, cpu_model, (d, vocab) = build(args, dev, rank, world)
    batches = make_batches(args, dev, rank, d, vocab)
    # --exchange peer: no collective and no fetch step for the codes -- the peers' shards are mapped into this process (HIP
    # IPC) and the star-attention / gather kernels read every row from its owner's memory themselves; only sharded LABELS
    # (--shard-vals) are still gathered into a local buffer (one kernel)
    fetcher = None
    if sharded and args.exchange == "peer":
        pf = PeerMappedFetcher(eng.store, shard, share_vals=args.shard_vals)
        eng.store = pf.mapped_store()
        eng.store.vals_row0 = shard.store_row0 if args.shard_vals else 0

        class _MappedCodes:                                       # the bench's fetcher interface over the mapped store
            link_bytes = 0

            def fetch_codes(self, ids, left, right, centres_only):
                n = ids.numel() if centres_only else ids.numel() * (1 + left + right)
                self.link_bytes += int(n * (world - 1) / world) * eng.store.codes.shape[1]     # rows the kernels pull over the links
                return None, None, None

            def account_merged(self, n_ids, n_groups, left, right):
                # multi-layer model on the mapped store: layer 0's star edges read every neighbour's centre row, the ntgt pipeline
                # the slots of the DISTINCT groups only (merged on the device before any row is touched)
                self.link_bytes += int((n_ids + n_groups * (1 + left + right)) * (world - 1) / world) * eng.store.codes.shape[1]

            def fetch_knn_vals(self, knn_ids):
                r = pf.fetch_knn_vals(knn_ids)
                self.link_bytes += pf.link_bytes
                pf.link_bytes = 0
                return r

            def check(self):
                pass
        fetcher = _MappedCodes()
    elif sharded:
        fetcher = ShardedFetcher(eng.store, shard, mode=args.exchange)
    centres_only = args.layers == 1
    # ---- --search device: the kNN index the step searches (one full replica per rank: 6.6 GB of codes + their tile image + the
    # payloads next to a 13-GB code store; what is range-sharded over the ranks is the GRAPH's store, BASELINE.json configs[2])
    search_mode = "given" if (args.small or args.graph or args.shard_vals or args.lmbda <= 0) else args.search
    idx = None
    if search_mode == "device":
        from gnnlm_amd.synthetic import synthetic_ivfpq_index
        idx = synthetic_ivfpq_index(args.n_store, eng.hgt.hidden_dim, 4096, 64, dev, nprobe=32)
        idx.attach_vals(eng.store.vals)                           # index key ids = store rows: payload = id << 24 | label
    # multi-layer model: the engine fetches inside the step, AFTER merging equal context groups -- one request per distinct centre
    # row of the batch (hgt.py; the one-layer step prefetches the next batch's centre rows on a side stream instead, below)
    fetch_in_step = fetcher is not None and args.layers > 1
    # batches in flight: also with a sharded store whose rows are prefetched on the fetch stream (the collectives stay on that one stream,
    # in program order on every rank; the lanes only carry the math and the search) -- so that N = 1 and N > 1 time the same step
    n_lanes = max(1, args.lanes) if (idx is not None and args.streams == 1 and not fetch_in_step) else 1
    if fetch_in_step and isinstance(fetcher, ShardedFetcher):
        eng.fetcher, eng.fetch_vals = fetcher, bool(args.shard_vals)
    merged_counts = []
    acc = torch.zeros(1, device=dev, dtype=torch.float64)

    assert args.blocks % args.streams == 0, "--blocks must be a multiple of --streams"
    side = [torch.cuda.Stream(device=dev) for _ in range(args.streams - 1)]
    accs = [acc] + [torch.zeros(1, device=dev, dtype=torch.float64) for _ in side]

    # Sharded store: the all-to-all row fetch of step i+1 runs on its own stream under the math of step i
    # (the inputs of every step are known up front); RCCL orders itself after the stream it is issued on.
    fetch_stream = torch.cuda.Stream(device=dev) if fetcher is not None else None
    # the math runs on a high-priority stream when there is an exchange to overlap: the exchange's kernels (bucketing,
    # owner-side gather, RCCL copies) then fill the tails of the compute kernels instead of time-slicing with them
    prio = int(os.environ.get("GNNLM_COMPUTE_PRIORITY", "-1"))
    compute_stream = torch.cuda.Stream(device=dev, priority=prio) if fetcher is not None and prio != 0 else None
    if compute_stream is not None:
        compute_stream.wait_stream(torch.cuda.current_stream())
        torch.cuda.set_stream(compute_stream)
    pending = {}

    n_fetches = [0]

    def issue_fetch(bi):
        n_fetches[0] += 1
        b = batches[bi % len(batches)]
        with torch.cuda.stream(fetch_stream):
            codes, valid, index = fetcher.fetch_codes(b.ids, 2, 2, centres_only)
            kv = fetcher.fetch_knn_vals(b.knn_ids) if args.shard_vals else None
            ev = torch.cuda.Event()
            ev.record(fetch_stream)
        pending[bi] = (codes, valid, index, kv, ev)

    def take_fetched(bi):
        """The rows batch bi asked for (sharded store, prefetched on the fetch stream) -> the batch, ordered before the current stream."""
        b = batches[bi % len(batches)]
        if bi not in pending:
            issue_fetch(bi)
        codes, valid, index, kv, ev = pending.pop(bi)
        cur = torch.cuda.current_stream()
        cur.wait_event(ev)
        for t in (codes, valid, index, kv):
            if t is not None:
                t.record_stream(cur)
        b.fetched_codes, b.fetched_valid, b.fetched_index, b.knn_vals = codes, valid, index, kv
        b.fetched_centres_only = centres_only
        return b

    def score_one(bi, a, prefetch_next):
        b = batches[bi % len(batches)]
        if fetch_in_step:
            n_fetches[0] += 1
            if not isinstance(fetcher, ShardedFetcher) and args.shard_vals:
                b.knn_vals = fetcher.fetch_knn_vals(b.knn_ids)
        elif fetcher is not None:
            b = take_fetched(bi)
        out = eng.score(b, args.lmbda, args.temperature, knn_index=idx, k=args.k)
        ops.masked_sum_f64(out["logp"], None, a)                  # score_sum (eval_lm.py:273)
        if fetch_in_step and len(merged_counts) < 64:
            merged_counts.append(eng.hgt._last_groups)            # (n, device counter) -- read after the run
        if fetcher is not None and not fetch_in_step and prefetch_next is not None:
            issue_fetch(prefetch_next)                            # host waits for the split sizes while the GPU computes

    # --search device with --lanes > 1: batch i runs on lane i % lanes (its own stream, its own accumulator); the lane's previous
    # batch is finished (the search's one host read, the interpolation) right before the lane's next batch is enqueued
    lane_streams = [torch.cuda.current_stream()] + [torch.cuda.Stream(device=dev) for _ in range(n_lanes - 1)]
    lane_accs = [acc] + [torch.zeros(1, device=dev, dtype=torch.float64) for _ in range(n_lanes - 1)]
    lane_pending = [None] * n_lanes
    for s_ in lane_streams[1:]:
        s_.wait_stream(lane_streams[0])

    def finish_lane(j):
        if lane_pending[j] is not None:
            with torch.cuda.stream(lane_streams[j]):
                out = eng.score_finish(lane_pending[j])
                ops.masked_sum_f64(out["logp"], None, lane_accs[j])
            lane_pending[j] = None

    def drain_lanes():
        for j in range(n_lanes):
            finish_lane(j)

    def step(i, last=False):
        if n_lanes > 1:
            j = i % n_lanes
            finish_lane(j)
            with torch.cuda.stream(lane_streams[j]):
                if fetcher is not None:
                    b = take_fetched(i)
                else:
                    b = batches[i % len(batches)]
                lane_pending[j] = eng.score_begin(b, args.lmbda, args.temperature, knn_index=idx, k=args.k)
            if fetcher is not None and not last:
                issue_fetch(i + 1)                                    # the next batch's rows: requested now, on the fetch stream
            return
        main = torch.cuda.current_stream()
        for s_i in range(args.streams):
            bi = i * args.streams + s_i
            nxt = None if (last or args.streams > 1) else bi + 1
            if s_i == 0:
                score_one(bi, accs[0], nxt)
            else:
                side[s_i - 1].wait_stream(main) if i == 0 else None
                with torch.cuda.stream(side[s_i - 1]):
                    score_one(bi, accs[s_i], None)

    def barrier():
        drain_lanes()                                             # (batches still in flight on the lanes belong to the steps before the barrier)
        torch.cuda.synchronize()                                  # all streams of this device
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    graphs = {}

    def capture_graphs():
        """One HIP graph per pooled batch: the ~45 launches of a step (all stream-ordered, device-side row
        counts, caller-provided workspaces: nothing in the step touches the host) replay as one submission."""
        assert fetcher is None and args.streams == 1, "--graph: single stream, no exchange"
        for bi in range(len(batches)):
            gph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gph):
                out = eng.score(batches[bi], args.lmbda, args.temperature)
                ops.masked_sum_f64(out["logp"], None, acc)
            graphs[bi] = gph

    # ---- parity gate FIRST (7 s of CPU-only oracle with the GPU idle: nothing timed may follow it closely): the engine that is
    # about to be timed against the oracle on a prefix of its first block
    if rank == 0:
        stage("parity gate")
    parity = None
    if rank == 0 and fetcher is None and not args.no_parity:
        parity = verify_block(eng, batches[0], args, cpu_model, min(args.tokens_per_sample, 256 if args.layers == 1 else 64))
    dlogp = None
    if args.precision != "f32" and fetcher is None:               # accuracy of the opt-in mode on one real batch
        got = eng.score(batches[0], args.lmbda, args.temperature)["logp"].clone()
        eng.hgt.gemm_precision = eng.asm.gemm_precision = 0
        ref = eng.score(batches[0], args.lmbda, args.temperature)["logp"]
        eng.hgt.gemm_precision = eng.asm.gemm_precision = ops.PRECISIONS[args.precision]
        dlogp = (got.double() - ref.double()).abs().max().item()
    # ---- warm-up: one plain step, then two profiled ones of which the second is kept (the first absorbs one-time costs that
    # would otherwise be charged to whatever kernel they happen beside: event / pinned-slot pools, profiler tool start-up)
    if rank == 0:
        stage("warm-up")
    step(0)
    barrier()
    search_gate = None
    if idx is not None and rank == 0 and args.search_check > 0 and not args.no_parity:
        # the search the step is about to run, against the float64 oracle on the step's own queries (refuses to go on otherwise)
        # (with a sharded store the HGT forward is a collective step: rank 0 alone checks the search on the block's input features)
        q0 = eng.features(batches[0]) if fetcher is None else batches[0].tgt_feats.float()
        search_gate = search_check(idx, (q0 / q0.norm(dim=1, keepdim=True)).contiguous(), args.k, args.search_check)
        del q0
    for i in (1, 2):
        _lib.profile_begin()
        step(i)
        drain_lanes()
        torch.cuda.synchronize()
        kern = _lib.profile_end()
    for i in range(max(0, args.warmup - 3)):
        step(i + 3)
    dominant = max(kern, key=lambda k_: kern[k_]["total_ms"])
    names = []
    while _lib.lib().gnnlm_kernel_name(len(names)) is not None:
        names.append(_lib.lib().gnnlm_kernel_name(len(names)).decode())
    if fetch_in_step and isinstance(fetcher, ShardedFetcher) and args.exchange == "padded" and eng.hgt._last_groups is not None:
        # merged requests in the fixed-capacity exchange: buckets sized from the distinct-group count the warm-up measured (agreed
        # among the ranks by a MAX all-reduce, here, outside the timed region) instead of the worst case
        fetcher.calibrate_groups(eng.hgt._last_groups[1])
    if args.graph:
        capture_graphs()
    # ---- settle (untimed): keep stepping back to back until at least `--settle-s` seconds of GPU work have run and two
    # successive chunks agree within 2 % -- the driver's 20-step window is 0.1 s long, and 0.1 s right after seconds of host-only
    # work is timed at whatever clock / power state the part is still ramping through (BENCH_r03: 5.58 ms against 4.73)
    if rank == 0:
        stage("settle")
    settle = {"seconds": 0.0, "steps": 0, "chunks_ms_per_step": []}
    chunk_n = max(10, min(args.steps, 50))
    nstep = [args.warmup]
    while settle["seconds"] < args.settle_max_s:
        torch.cuda.synchronize()
        tc = time.perf_counter()
        for _ in range(chunk_n):
            if args.graph:
                graphs[nstep[0] % len(batches)].replay()
            else:
                step(nstep[0])
            nstep[0] += 1
        settle["sclk_mhz_under_load"] = read_sclk(local_rank, sysfs_only=True)     # the chunk is still running on the device
        drain_lanes()
        torch.cuda.synchronize()
        el = time.perf_counter() - tc
        if world > 1:                                             # every rank must take the same decision: the steps hold collectives
            tel = torch.tensor([el], device=dev, dtype=torch.float64)
            all_reduce(tel, dist.ReduceOp.MAX)
            el = tel.item()
        settle["seconds"] += el
        settle["steps"] += chunk_n
        settle["chunks_ms_per_step"].append(round(el / chunk_n * 1e3, 4))
        c = settle["chunks_ms_per_step"]
        if settle["seconds"] >= args.settle_s and len(c) >= 2 and abs(c[-1] - c[-2]) <= 0.02 * c[-1]:
            break
    settle["seconds"] = round(settle["seconds"], 3)
    # ---- timed region: exactly K steps, the dominant kernel bracketed by HIP events on its stream; one more event per step
    # boundary on the same stream gives the per-step times (median / min / max beside the wall-clock mean)
    for a in accs + lane_accs:
        a.zero_()
    pending.clear()                                               # nothing fetched ahead of the timed region
    if rank == 0:
        stage("timed region")
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    barrier()
    if args.graph:
        t0 = time.perf_counter()
        marks[0].record()
        for i in range(args.steps):
            graphs[i % len(batches)].replay()
            marks[i + 1].record()
        sclk_timed = read_sclk(local_rank, sysfs_only=True)
        barrier()
        dt = time.perf_counter() - t0
        prof = dict(kern[dominant])                               # from the profiled warm-up step (see --graph)
        prof["launches"] *= args.steps; prof["total_ms"] *= args.steps; prof["flops"] *= args.steps; prof["bytes"] *= args.steps
    else:
        _lib.profile_begin(1 << names.index(dominant))
        t0 = time.perf_counter()
        marks[0].record()
        for i in range(args.steps):
            step(i, last=(i == args.steps - 1))
            marks[i + 1].record()
        sclk_timed = read_sclk(local_rank, sysfs_only=True)       # host ahead of the device: the clock of the timed steps (~0.1 ms of host time)
        barrier()
        dt = time.perf_counter() - t0
        prof = _lib.profile_end()[dominant]
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)) if args.streams == 1 and n_lanes == 1 else []
    for a in accs[1:] + lane_accs[1:]:
        acc += a
    # ---- the same K steps with the search results GIVEN (the batches' precomputed sims / ids: rounds 1-5 quoted this as `value`)
    if rank == 0:
        stage("search-given leg")
    given = None
    search_stats = None
    if idx is not None:
        search_stats = {k_: (float(v_.item()) if torch.is_tensor(v_) else v_) for k_, v_ in idx.stats.items()}
        if fetcher is None:
            acc_g = torch.zeros(1, device=dev, dtype=torch.float64)
            run_g = lambda i: ops.masked_sum_f64(eng.score(batches[i % len(batches)], args.lmbda, args.temperature)["logp"], None, acc_g)
            for i in range(3):
                run_g(i)
            barrier()
            t0g = time.perf_counter()
            for i in range(args.steps):
                run_g(i)
            barrier()
            dtg = time.perf_counter() - t0g
            if world > 1:
                ttg = torch.tensor([dtg], device=dev, dtype=torch.float64)
                all_reduce(ttg, dist.ReduceOp.MAX)
                dtg = ttg.item()
            given = {"tokens_per_s": round(args.steps * args.blocks * args.tokens_per_sample * world / dtg, 1), "ms_per_step": round(dtg / args.steps * 1e3, 4)}
    link_bytes_per_step = replicated = None
    merged_groups = None
    if fetcher is not None:
        fetcher.check()                                           # no request was dropped by the fixed-capacity buckets
        mc = [(g_[0], int(g_[1][0].item())) for g_ in merged_counts if g_ is not None]
        if mc:
            merged_groups = {"context_groups_per_step": mc[0][0], "distinct_requested_per_step_mean": round(sum(c_ for _, c_ in mc) / len(mc), 1)}
            if not isinstance(fetcher, ShardedFetcher):           # mapped store: no fetch call to count bytes in
                fetcher.link_bytes = 0
                for n_, c_ in mc:
                    fetcher.account_merged(n_, c_, 2, 2)
                fetcher.link_bytes *= n_fetches[0] / len(mc)
        link_bytes_per_step = fetcher.link_bytes / max(1, n_fetches[0])
        # the same steps on a REPLICATED store (every rank holds the whole table, no exchange): what the sharding costs
        from gnnlm_amd.engine import GnnLmEngine
        from gnnlm_amd.hgt import CodeStore
        from gnnlm_amd.synthetic import device_codes
        st = eng.store
        full = st.codes if st.codes.shape[0] == args.n_store else device_codes(args.n_store, st.codes.shape[1], dev, 1234)
        vfull = st.vals if st.vals.shape[0] == args.n_store else None
        if vfull is not None:
            eng_r = GnnLmEngine(eng.hgt, eng.asm, CodeStore(codes=full, centroids=st.centroids, n_store=args.n_store, row0=0,
                                                            vals=vfull, A=st.A, b=st.b), 2, 2)
            for b_ in batches:
                b_.fetched_codes = b_.fetched_valid = b_.fetched_index = b_.knn_vals = None
                b_.fetched_centres_only = False
            acc_r = torch.zeros(1, device=dev, dtype=torch.float64)
            run_r = lambda i: ops.masked_sum_f64(eng_r.score(batches[i % len(batches)], args.lmbda, args.temperature, knn_index=idx, k=args.k)["logp"], None, acc_r)   # (the same step: search inside, one batch at a time)
            run_r(0)
            barrier()
            _lib.profile_begin()                                  # as in the timed region above: the per-launch events cost ~2 % of a step
            t0r = time.perf_counter()
            for i in range(args.steps):
                run_r(i)
            barrier()
            dtr = time.perf_counter() - t0r
            _lib.profile_end()
            if world > 1:
                ttr = torch.tensor([dtr], device=dev, dtype=torch.float64)
                all_reduce(ttr, dist.ReduceOp.MAX)
                dtr = ttr.item()
            replicated = {"tokens_per_s": round(args.steps * args.blocks * args.tokens_per_sample * world / dtr, 1),
                          "ms_per_step": round(dtr / args.steps * 1e3, 4)}
    if rank == 0:
        stage("extras")
    recipe = drv = search = None
    if rank == 0 and world == 1 and fetcher is None and not args.small and args.extras:
        if args.layers == 1 and args.precision == "f32":
            stage("extras: recipe_L3")
            recipe = recipe_l3(args, eng, batches, dev)
        stage("extras: driver_path")
        drv = driver_path(args, eng, batches, dev)
        stage("extras: knn_search")
        l3_ids = recipe.pop("_l3_ids", None) if recipe is not None else None
        if args.precision == "f32":
            search = knn_search(args, eng, batches, dev, dt / args.steps * 1e3, l3_ids, idx=idx)
        del l3_ids
    if recipe is not None and search is not None and recipe["tokens_per_step"] == 4096:
        # step + on-device search of the step's own queries (what the reference's timer spans), for the recipe's lines
        for key in ("uniform_ids",):
            e = recipe[key]
            e["tokens_per_s_with_search"] = round(recipe["tokens_per_step"] / ((e["ms_per_step"] + search["ms_per_4096_queries"]) / 1e3), 1)
        w = recipe["searched_neighbours"]["within_batch_merge"]
        w["tokens_per_s_with_search"] = round(recipe["tokens_per_step"] / ((w["ms_per_step"] + search["ms_per_4096_queries"]) / 1e3), 1)
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        all_reduce(tt, dist.ReduceOp.MAX)
        dt = tt.item()
        all_reduce(acc)
    tokens = args.steps * args.blocks * args.tokens_per_sample * world
    score_sum = acc.item()

    def knn_interp_ab():
        """The label gather of the ids-only interpolation both ways, on the first batch: the one-pass kernel (one memory request per
        look-up: the memory system serves ~48-51 G requests/s whatever their size, tools/probes/fetch_calib.hip -- FETCH_SIZE counts
        exactly these requests) and the routed look-ups the step runs (csrc/knn_bucket.hip: sorted by region of the one-byte tag
        table, looked up against the region's slice in LDS)."""
        b0 = batches[0]
        if b0.knn_vals is not None or b0.knn_sims is None or fetcher is not None:
            return None
        st = eng.store
        lm0 = torch.zeros(b0.targets.shape[0], device=dev)
        out = {}
        for nm, kw in (("one_pass_us", dict(bucketed=False)), ("routed_us", dict(bucketed="auto"))):
            f = lambda: ops.knn_interp(lm0, b0.knn_sims, b0.knn_ids, b0.targets, args.temperature, args.lmbda, vals=st.vals, n_store=st.n_store,
                                       row0=getattr(st, "vals_row0", st.row0), **kw)
            f()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                f()
            e1.record()
            torch.cuda.synchronize()
            out[nm] = round(e0.elapsed_time(e1) * 100, 1)
        n_tok = b0.targets.shape[0]
        req = n_tok * args.k * (1 + 12 / 128.0)                       # label gathers + the id / similarity stream in 128-byte requests
        out["one_pass_requests_per_s_G"] = round(req / (out["one_pass_us"] * 1e-6) / 1e9, 1)
        out["one_pass_frac_of_measured_request_ceiling_48G"] = round(req / (out["one_pass_us"] * 1e-6) / 48e9, 3)
        out["note"] = "back-to-back calls with warm caches; inside the step (`kernels[].avg_us`) both are ~15 % slower"
        return out

    def roof(name, e):
        if name == "ivfpq_scan8_kernel" and search_stats is not None:
            return filter_roofline(e["total_ms"], e["launches"], search_stats["pairs"] * e["launches"], pmc_traffic(name)[0])
        r_ = roofline_entry(name, e, args.precision)
        if name == "knn_interp_kernel" and knn_ab is not None:
            r_["label_gather"] = knn_ab
        return r_
    knn_ab = knn_interp_ab() if rank == 0 else None
    if rank == 0:
        r = roof(dominant, prof)      # (the filter: `pairs` of idx.stats are the (query, key) pairs of ONE search -- every launch of the timed region scans as many)
        r["traffic"], r["traffic_source"] = pmc_traffic(dominant)
        r["launches_per_step"] = prof["launches"] / args.steps
        res = {
            "metric": "eval tokens/sec on WikiText-103 (k=1024, GNN+KNN); test ppl match",
            "value": round(tokens / dt, 1), "unit": "tokens/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "ms_per_step_median": (round(per_step[len(per_step) // 2], 4) if per_step else None),
            "ms_per_step_min": (round(per_step[0], 4) if per_step else None),
            "ms_per_step_max": (round(per_step[-1], 4) if per_step else None),
            "settle": settle, "sclk_mhz_timed_region": sclk_timed,
            "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "value_includes_search": idx is not None,
            "value_search_given": (given["tokens_per_s"] if given is not None else (round(tokens / dt, 1) if idx is None else None)),
            "ms_per_step_search_given": (given["ms_per_step"] if given is not None else None),
            "dtype": "f32" if args.precision == "f32" else f"f32 via {args.precision} split-bf16 MFMA", "data": "synthetic",
            "config": {"workload": f"BASELINE.json configs[1]: WikiText-103 full PQ datastore in HBM, k_g={args.gcn_k}, "
                                   f"context 2+2, HGT {args.layers} layer{'s' if args.layers > 1 else ''}, kNN k={args.k} "
                                   + (f"(`value`: the step WITH the on-device IVF-PQ search of its own queries = what the reference's own timer spans, "
                                      f"fairseq_cli/eval_lm.py:214-219 -> knn_model.py:100; `value_search_given`: the same steps with precomputed search "
                                      f"results, SURVEY.md 8d), " if idx is not None else "(`value`: search results given, SURVEY.md 8d), ")
                                   + f"{args.tokens_per_sample}-token blocks",
                       "n_store": args.n_store, "blocks_per_step_per_gpu": args.blocks, "streams": args.streams, "hip_graph": bool(args.graph), "tokens_per_block": args.tokens_per_sample,
                       "gcn_k": args.gcn_k, "knn_k": args.k, "hgt_layers": args.layers, "d": d, "vocab": vocab, "neighbour_ids": args.ids,
                       "lmbda": args.lmbda, "temperature": args.temperature,
                       "knn_search": ({"where": "on the device, inside the timed step", "index": "synthetic OPQ64_1024,IVF4096,PQ64 (one replica per rank)",
                                       "keys": args.n_store, "nprobe": 32, "k": args.k, "lanes": n_lanes,
                                       "pairs_per_query": round(search_stats["pairs"] / max(1, search_stats["queries"])),
                                       "survivors_per_query": round(search_stats["survivors"] / max(1, search_stats["queries"])),
                                       "queries_searched_again_last_step": search_stats.get("requeried", 0), "parity": search_gate}
                                      if idx is not None else {"where": "results given with the batch"}),
                       "store": (("range-sharded + RCCL all-to-all" if transport == "nccl" else f"range-sharded + host-staged {transport} all-to-all (test transport: no RCCL, no xGMI)")
                                 if sharded else "replicated" if world > 1 else "single GPU"),
                       "collective_backend": transport,
                       "rccl_ranks": world if transport == "nccl" else 0,
                       "exchange": (args.exchange if sharded else None),
                       "exchange_requests": ((("one per DISTINCT context group of the batch (merged on the device before the exchange; halo layout)" if args.layers > 1 else "centre rows only")) if sharded else None),
                       "merged_groups": merged_groups,
                       **({"xgmi_bytes_per_step_per_rank": round(link_bytes_per_step)} if (link_bytes_per_step is not None and transport == "nccl") else
                          {"exchange_bytes_per_step_per_rank_host_staged": round(link_bytes_per_step)} if link_bytes_per_step is not None else {}),
                       "replicated_store": replicated,
                       "gemm_precision": args.precision, "max_abs_dlogp_vs_f32": dlogp,
                       "centre_state_cache": "off in the timed region (a cycled pool would be all hits); see recipe_L3 for the cached lines",
                       "synthetic_ppl": round(float(2 ** (-score_sum / tokens / np.log(2))), 4)},
            "roofline": r,
            "kernels": [roof(k_, v) for k_, v in sorted(kern.items(), key=lambda kv: -kv[1]["total_ms"])],
        }
        if parity is not None:
            res["parity"] = parity
        if recipe is not None:
            res["recipe_L3"] = recipe
        if drv is not None:
            res["driver_path"] = drv
        if search is not None:
            res["knn_search"] = search
        if world == 1 and not args.no_cpu_baseline:
            stage("cpu_baseline")
            res["cpu_baseline"] = cpu_baseline(args, cpu_model)
        print(json.dumps(res))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
