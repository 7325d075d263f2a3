"""-m gpu: "identical top-k doc IDs, cosine scores within 1e-3" (north_star) at a size where it means something — the
encode -> retrieve chain through the drop-in entry points over 512 structured synthetic pages + the reference's OWN four
input images (cat.jpeg 3x3 slices, dog.jpg 2x2, the two InfoVQA pages of examples/training_data/0.parquet: byte copies
under tests/golden/inputs/) x 512 synthetic queries + the two parquet queries, top-10, against what the REFERENCE returned
for the same inputs (tests/golden/config1xl_full.npz, oracle/gen_golden.py --config1xl: 19 CPU-minutes of openmatch's
DRModelForInference + distributed_parallel_retrieve):

  * every embedding cosine >= 1 - 1e-3, every one of the 514 x 516 query x document scores within 1e-3;
  * IDENTICAL top-10 id sets for the 33 queries whose reference rank-10 / rank-11 gap exceeds 2e-3, tolerance-equivalent sets
    and tolerance-consistent order for all 514;
  * the unconditional numbers (ids identical in order, overlap@10) are recorded in gpurun_out/config1xl_parity.json and by
    bench.py's `reference_parity` block."""
import json
import os
import types

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from tests import config1xl_util as X  # noqa: E402
from visrag_amd import utils as U  # noqa: E402
from visrag_amd.config import full_config  # noqa: E402
from visrag_amd.inference import distributed_parallel_embedding_inference  # noqa: E402
from visrag_amd.modeling import DRModelForInference  # noqa: E402
from visrag_amd.retriever import distributed_parallel_retrieve  # noqa: E402
from visrag_amd.synth import iter_synth_weights  # noqa: E402
from visrag_amd.tokenizer import StandInTokenizer  # noqa: E402


def test_config1xl_encode_retrieve_top10(tmp_path):
    g, man = X.load_fixture()
    cfg = full_config()
    model = DRModelForInference.build(cfg=cfg, state_dict=iter_synth_weights(cfg, 0, device="cuda"), max_images=48,
                                      max_tokens=8192, max_seqs=64, pipeline=2)
    try:
        tok = StandInTokenizer(cfg.vocab_size)
        corpus, queries = X.corpus_and_queries(g, man)
        args = types.SimpleNamespace(output_dir=str(tmp_path), per_device_eval_batch_size=32, process_index=0, world_size=1,
                                     max_inmem_docs=129, device=f"cuda:{model.encoder.device}")
        distributed_parallel_embedding_inference(corpus, model, args, "corpus", True, {"tokenizer": tok, "max_inp_length": 2048})
        distributed_parallel_embedding_inference(queries, model, args, "query", False, {"tokenizer": tok, "max_inp_length": 512})
        shards = U.list_shards(str(tmp_path), "corpus")
        assert len(shards) == 4                                       # 516 documents in files of 129: the reference's 4 shards
        P = np.concatenate([U.read_shard(p)[0] for p in shards])
        ids = [i for p in shards for i in U.read_shard(p)[1]]
        Q, qids = U.read_shard(U.list_shards(str(tmp_path), "query", 0)[0])
        assert ids == [str(x) for x in g["doc_ids"]] and qids == [f"q{i}" for i in range(len(Q))]
        run = distributed_parallel_retrieve(args, int(g["k"]))          # reference semantics: union of the per-shard top-k
        st = X.parity_stats(g, P, Q, run)
        print(json.dumps(st, indent=1))
        out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "gpurun_out")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "config1xl_parity.json"), "w") as f:
            json.dump(st, f, indent=1)
        X.assert_bars(st)
    finally:
        for enc, _ in model._slots:
            enc.close()
