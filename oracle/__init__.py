"""CPU oracle for the GNN+kNN eval hot path of ShannonAI/GNN-LM.

THIS PACKAGE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.

It is a plain numpy / torch-CPU restatement of the reference algorithm *as
written* (every function cites the reference file:line it follows).  Only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it -- as the checker / the timed CPU baseline, never as
something the product path (``gnnlm_amd``) routes through.  The product path
fails loudly when the HIP library is missing; it never falls back to this code.

Parity pinning
--------------
The reference holds no test for this path, so the oracle is
pinned against *outputs of the reference itself run in the authoring
container*: ``tests/golden/make_golden.py`` imports the reference's own
``knn/pq_wrapper.py``, ``knn/knn_model.py``, ``knn/data_store.py``,
``fairseq/models/hgt.py`` (under a pure-torch stand-in for the absent ``dgl``
package) and the static edge builders of ``GraphTokenBlockDataset`` and stores
inputs + outputs as ``tests/golden/*.npz``.  ``tests/test_oracle_golden.py``
checks every oracle function against those vectors.

Two third-party boundaries stay **parity unpinned** (libraries absent from the
image, no reference test pins them): faiss ANN search results / ADC distances
(``knn_model.py:100``) and real-DGL corner semantics (zero in-degree reducers).
"""
