"""Generates tests/golden/launcher.json by importing the REFERENCE's legion_server.py (runs only in
the build container, where /root/reference exists; the fixture it writes is data: inputs and the
reference's outputs)."""
import importlib.util
import json
import math
import os
import types

import networkx as nx

REF = "/root/reference/legion_server.py"
spec = importlib.util.spec_from_file_location("ref_legion_server", REF)
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

out = {"meta_config": [], "topo": []}
for name in ["products", "paper100m", "com-friendster", "ukunion", "uk2014", "clueweb"]:
    for bs, cm, ep, path in [(8000, 38000000, 2, "dataset"), (1024, 13000000000, 10, "/data/legion")]:
        args = types.SimpleNamespace(dataset_name=name, dataset_path=path, train_batch_size=bs, fanout=[25, 10],
                                     gpu_number=1, epoch=ep, cache_memory=cm, usenvlink=0)
        cwd = os.getcwd()
        os.chdir("/tmp")
        saved = ref.os.system
        ref.os.system = lambda cmd: 0          # do not launch the (absent) binary
        try:
            ref.Run(args)
        finally:
            ref.os.system = saved
            line = open("/tmp/meta_config").read()
            os.chdir(cwd)
        out["meta_config"].append({"dataset_name": name, "dataset_path": path, "train_batch_size": bs,
                                   "cache_memory": cm, "epoch": ep, "line": line})


def matrix(n, linked):
    hdr = "\t" + "\t".join(f"GPU{i}" for i in range(n)) + "\tCPU Affinity"
    rows = [hdr]
    for i in range(n):
        cells = []
        for j in range(n):
            cells.append("X" if i == j else ("NV12" if linked(i, j) else "SYS"))
        rows.append(f"GPU{i}\t" + "\t".join(cells) + "\t0-63")
    return "\n".join(rows) + "\n\nLegend:\n  X = Self\n  NV# = Connection traversing a bonded set of # NVLinks\n"


cases = {"dgx_a100_full_mesh_8": matrix(8, lambda i, j: True),
         "siton_pairs_8": matrix(8, lambda i, j: i // 2 == j // 2),
         "dgx_v100_quads_8": matrix(8, lambda i, j: i // 4 == j // 4),
         "ring_4": matrix(4, lambda i, j: (i - j) % 4 in (1, 3)),
         "no_links_2": matrix(2, lambda i, j: False),
         "single_gpu": matrix(1, lambda i, j: False)}
for name, text in cases.items():
    conns = ref.parse_topo_output(text)
    G = nx.Graph()
    G.add_edges_from(conns)
    size, _ = ref.find_largest_fully_connected_group(G)
    out["topo"].append({"name": name, "text": text, "connections": conns, "clique_size": size,
                        "cache_agg_mode": math.log2(size)})
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "launcher.json"), "w"), indent=1)
print("meta_config cases:", len(out["meta_config"]), "topo cases:", len(out["topo"]))
