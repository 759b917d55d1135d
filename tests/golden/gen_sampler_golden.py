"""Regenerates tests/golden/sampler_tiny.json with the CPU oracle (the reference ships no vectors,
SURVEY.md F4: these known answers are this repository's own, cross-checked in
tests/test_oracle_sampler.py against an independent pure-Python restatement)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.test_oracle_sampler import oracle_batch, tiny_graph  # noqa: E402

indptr, col = tiny_graph()
cases = []
for seeds, fanout, bs, counter in [([0, 6, 7, 3], [3, 2], 4, 0), ([0, 1, 2, 3, 4, 5, 6, 7], [25, 10], 8, 0),
                                   ([5, 4, 3, 2, 1, 0], [2, 2, 2], 3, 1), ([7, 6], [4], 2, 0)]:
    out = oracle_batch(indptr, col, np.array(seeds, dtype=np.int32), fanout, bs, counter)
    cases.append({"seeds": seeds, "fanout": fanout, "batch_size": bs, "counter": counter,
                  **{k: out[k].tolist() for k in ("sampled_ids", "agg_src_off", "agg_dst_off", "node_counter",
                                                  "edge_counter")}})
json.dump({"graph": {"indptr": indptr.tolist(), "col": col.tolist()}, "cases": cases},
          open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "sampler_tiny.json"), "w"), indent=1)
print("wrote", len(cases), "cases")
