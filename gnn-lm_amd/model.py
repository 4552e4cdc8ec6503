"""``GnnLmModel`` -- the eval-path behaviour of ``TransformerLanguageModel`` with a
``TokenGraphTransformerDecoder`` (fairseq/models/transformer.py:910-1085) under
``--use-precompute-feat``: the base LM is bypassed (:974-976), the graph decoder (HGT) refines the
precomputed features, and the tied adaptive softmax scores the targets.

State-dict names follow the reference (prefix ``decoder.``): ``hgt_decoder.gcs.*``,
``tgt_quantizer.*`` (convert_ckpt.py:40-45), ``embed_tokens.embeddings.{i}.{0,1}.weight``,
``adaptive_softmax.head.class_proj.weight``.
"""
import os
from argparse import Namespace

import torch

from ._lib import raw_stream as _lib_raw_stream
from .adaptive_softmax import AdaptiveSoftmax
from .hgt import HGT, CodeStore, NeighborGraph
from .pq_wrapper import TorchPQCodec


class GnnLmModel(torch.nn.Module):
    graph_capture = False              # (class defaults: subclasses that script forward() need not call __init__)
    _graphs, _static_x = None, frozenset()

    def __init__(self, hgt: HGT, asm: AdaptiveSoftmax, quantizer: TorchPQCodec = None, orig_prob_ratio: float = 0.0,
                 short_cut: bool = False):
        super().__init__()
        if orig_prob_ratio > 0:
            raise NotImplementedError("orig_prob_ratio > 0 needs the base-LM logits (transformer.py:987-1005); "
                                      "the GNN-LM recipes evaluate with 0.0")
        self.hgt_decoder, self.adaptive_softmax, self.tgt_quantizer = hgt, asm, quantizer
        self.short_cut = short_cut
        # `eval_lm --graph-capture`: the launches of forward() and of target_log_probs() replayed from HIP graphs, one pair per
        # batch shape (the recipe's literal one-block batches are launch-bound: ~45 launches for 0.2 ms of GPU work)
        self.graph_capture = False

    def eval(self):
        return self

    def forward(self, src_tokens, src_lengths=None, graph: NeighborGraph = None, **unused):
        """-> (x [bsz, tgt_len, d], extra) like TokenGraphTransformerDecoder.forward (:943-1009)."""
        if self.graph_capture and graph is not None and graph.tgt_h is not None and graph.fetcher is None \
                and graph.fetched_codes is None and not torch.cuda.is_current_stream_capturing():
            return self._forward_replayed(src_tokens, graph)
        return self._forward(src_tokens, graph)

    def _forward_replayed(self, src_tokens, graph):
        """The same call through a HIP graph captured once per batch shape: the batch's neighbour ids and features are copied into
        the graph's static input buffers (two small copies), one replay enqueues every launch of the step.  The outputs are the
        graph's static buffers -- valid until the next call of this shape (the scorer consumes them at once)."""
        import dataclasses
        key = ("fwd", tuple(src_tokens.shape), tuple(graph.ids.shape), graph.tgt_h.dtype, graph.left, graph.right, graph.max_intra_context,
               id(graph.store), graph.store.codes.data_ptr(), _lib_raw_stream())      # (one set of static buffers per stream)
        if self._graphs is None:
            self._graphs, self._static_x = {}, set()
        e = self._graphs.get(key)
        if e is None:
            ids, feats = graph.ids.clone(), graph.tgt_h.clone()
            static = dataclasses.replace(graph, ids=ids, tgt_h=feats)
            self._forward(src_tokens, static)                     # eager once: one-time allocations happen outside the capture
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            # captured ON THE LANE'S OWN STREAM: the workspaces of the HGT and of the softmax are keyed by the stream the kernels are
            # enqueued on (hgt.py, adaptive_softmax.py) -- torch's default capture stream is one stream for every capture, i.e. every
            # lane's graph would get the SAME scratch addresses and the graphs, replayed concurrently on different lanes, would race
            cap = self._capture_stream()
            with torch.cuda.graph(g, stream=cap):
                out = self._forward(src_tokens, static)
            e = self._graphs[key] = {"graph": g, "ids": ids, "feats": feats, "out": out, "static": static, "capture_stream": cap}
            self._static_x.add(out[0].data_ptr())
        e["ids"].copy_(graph.ids)
        e["feats"].copy_(graph.tgt_h)
        e["graph"].replay()
        return e["out"]

    def release_stream_state(self, keep=()):
        """Drop what is held per stream for every stream but `keep` (raw handles): the HIP graphs captured for those lanes FIRST (their
        launches have the lanes' workspace addresses baked in), then the workspaces and merge tables themselves
        (HGT.release_stream_state).  Called by a driver whose side lanes are gone (eval_lm.main); the caller has joined the lanes."""
        keep = set(keep)
        if self._graphs:
            torch.cuda.synchronize()
            gone = [k_ for k_ in self._graphs if k_[0] == "fwd" and k_[-1] not in keep]
            xs = {self._graphs[k_]["out"][0].data_ptr() for k_ in gone}
            gone += [k_ for k_ in self._graphs if k_[0] == "asm" and k_[1] in xs]
            for k_ in gone:
                del self._graphs[k_]
            self._static_x = set(self._static_x) - xs
        for mod in (self.hgt_decoder, self.adaptive_softmax):
            if hasattr(mod, "release_stream_state"):
                mod.release_stream_state(keep=keep)

    @staticmethod
    def _capture_stream():
        """The stream a lane's graphs are captured on: the lane's own stream -- or, for the lane that runs on the default stream
        (which cannot capture), a private stream of that lane: either way no two lanes' captures see the same stream handle."""
        cur = torch.cuda.current_stream()
        return torch.cuda.Stream(device=cur.device) if cur == torch.cuda.default_stream(cur.device) else cur

    def _forward(self, src_tokens, graph):
        bsz, tgt_len = src_tokens.shape
        if graph is None or graph.tgt_h is None:
            raise ValueError("graph.tgt_h (precomputed tgt features) is required: the base LM is not built "
                             "(--use-precompute-feat path, transformer.py:974-976)")
        assert graph.n_blocks == bsz and graph.T == tgt_len
        h = graph.tgt_h
        if h.dtype != torch.float32:
            from . import ops
            h = ops.half_to_float(h.contiguous()) if h.dtype == torch.float16 else h.float()
        extra = {"inner_states": [h.view(bsz, tgt_len, -1).transpose(0, 1)]}
        x = h if self.short_cut else self.hgt_decoder(graph, features={"tgt": h})["tgt"]
        x = x.view(bsz, tgt_len, -1)
        extra["gcn_feat"] = x.transpose(0, 1)                               # :997
        return x, extra

    def target_log_probs(self, net_output, target):
        """log p(target) [bsz, tgt_len]: get_normalized_probs(log_probs=True) + gather
        (transformer.py:1064-1079, sequence_scorer.py:48-53,89) without the dense [T, V] tensor."""
        x = net_output[0]
        bsz, T, d = x.shape
        if self.graph_capture and x.data_ptr() in self._static_x and not torch.cuda.is_current_stream_capturing():
            key = ("asm", x.data_ptr(), bsz, T, d)                # x is a replayed forward's static output: its address names the shape's graph
            e = self._graphs.get(key)
            if e is None:
                tgt = target.reshape(-1).clone()
                self.adaptive_softmax.target_log_prob(x.reshape(-1, d).contiguous(), tgt)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                cap = self._capture_stream()
                with torch.cuda.graph(g, stream=cap):                          # (the lane's own stream: see _forward_replayed)
                    out = self.adaptive_softmax.target_log_prob(x.reshape(-1, d).contiguous(), tgt).view(bsz, T)
                e = self._graphs[key] = {"graph": g, "target": tgt, "out": out, "x": x, "capture_stream": cap}
            e["target"].copy_(target.reshape(-1))
            e["graph"].replay()
            return e["out"]
        return self.adaptive_softmax.target_log_prob(x.reshape(-1, d).contiguous(), target.reshape(-1)).view(bsz, T)

    def get_normalized_probs(self, net_output, log_probs, sample):
        raise NotImplementedError("the dense [B, T, V] tensor is never materialised; use target_log_probs()")

    @classmethod
    def from_checkpoint(cls, path, device, overrides=None, vocab_size=None, quantizer=None):
        """Load a reference ``.pt`` ({'args': Namespace, 'model': state_dict}, checkpoint_utils.py:161-176)."""
        ckpt = torch.load(path, map_location="cpu", weights_only=False)
        args = ckpt["args"] if isinstance(ckpt["args"], Namespace) else Namespace(**ckpt["args"])
        for k_, v in (overrides or {}).items():
            setattr(args, k_, v)
        sd = ckpt["model"]
        d, H, L = args.decoder_embed_dim, args.decoder_attention_heads, args.graph_layer
        hgt = HGT(in_dim=d, hidden_dim=getattr(args, "decoder_gcn_dim", d), out_dim=d, n_layers=L, n_heads=H)
        hgt.load_state_dict({k_[len("decoder.hgt_decoder."):]: v for k_, v in sd.items()
                             if k_.startswith("decoder.hgt_decoder.")})
        if quantizer is None:
            if "decoder.tgt_quantizer.centroids_torch" in sd:
                quantizer = TorchPQCodec.from_arrays(sd["decoder.tgt_quantizer.centroids_torch"].numpy(),
                                                     sd["decoder.tgt_quantizer.A"].numpy() if "decoder.tgt_quantizer.A" in sd else None,
                                                     sd["decoder.tgt_quantizer.b"].numpy() if "decoder.tgt_quantizer.b" in sd else None)
            elif getattr(args, "quantizer_path", "") and os.path.exists(str(args.quantizer_path)):
                quantizer = TorchPQCodec.from_file(args.quantizer_path)      # .npz or the recipe's faiss `quantizer` file
            else:
                raise ValueError("no quantizer: the checkpoint has no decoder.tgt_quantizer.* buffers "
                                 "(fairseq_cli/convert_ckpt.py) and quantizer_path does not name a file")
        cut = [int(c) for c in str(args.adaptive_softmax_cutoff).split(",")]
        if vocab_size is None:
            n_bands = sd["decoder.adaptive_softmax.head.class_proj.weight"].shape[0] + 1
            vocab_size = cut[0] + sum(sd[f"decoder.embed_tokens.embeddings.{i}.0.weight"].shape[0] for i in range(1, n_bands))
        asm = AdaptiveSoftmax.from_state_dict(sd, cut, vocab_size, device)
        return cls(hgt, asm, quantizer, getattr(args, "orig_prob_ratio", 0.0), getattr(args, "short_cut", False)), args

    def make_store(self, codes, n_store, device, vals=None, row0=0) -> CodeStore:
        q = self.tgt_quantizer
        t = lambda a: a.to(device).contiguous()
        return CodeStore(codes=codes, centroids=t(q.centroids_torch), n_store=n_store, row0=row0, vals=vals,
                         A=t(q.A) if q.pre_torch else None, b=t(q.b) if q.pre_torch and q.b.numel() else None)
