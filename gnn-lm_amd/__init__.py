"""gnnlm_amd -- MI355X-native implementation of the GNN+kNN eval hot path of ShannonAI/GNN-LM.

Host-side mirror of the reference's Python operator surface (DataStore, TorchPQCodec, HGT,
KNNModel, SequenceScorer, eval_lm) over the C ABI of libgnnlm_hip.so (include/gnnlm.h).
"""
__version__ = "0.1.0"
