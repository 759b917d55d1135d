"""Launcher for the MI355X sampling server: same CLI, same ./meta_config line and same
`<binary> <gpu_number> <cache_agg_mode>` hand-off as the reference's legion_server.py
(legion_server.py:39-127), with the GPU-link topology read from rocm-smi (xGMI) instead of
`nvidia-smi topo -m` (NVLink).  The dataset table is data copied from legion_server.py:41-88.
"""
import argparse
import math
import os
import subprocess

# name -> (directory, vertices, edges, feature dim, train, valid, test)   legion_server.py:41-88
DATASETS = {
    "products": ("products", 2449029, 123718280, 100, 196615, 39323, 2213091),
    "paper100m": ("paper100M", 111059956, 1615685872, 128, 11105995, 100000, 100000),
    "com-friendster": ("com-friendster", 65608366, 1806067135, 256, 6560836, 100000, 100000),
    "ukunion": ("ukunion", 133633040, 5507679822, 256, 13363304, 100000, 100000),
    "uk2014": ("uk2014", 787801471, 47284178505, 128, 78780147, 100000, 100000),
    "clueweb": ("clueweb", 955207488, 42574107469, 128, 95520748, 100000, 100000),
}


def meta_config_line(dataset_path, dataset_name, train_batch_size, cache_memory, epoch):
    """The single line of ./meta_config (legion_server.py:94-95 <-> storage_management.cu:29-61)."""
    d, n, e, f, tr, va, te = DATASETS[dataset_name]
    path = dataset_path + "/" + d + "/"
    return "{} {} {} {} {} {} {} {} {} {}".format(path, train_batch_size, n, e, f, tr, va, te, cache_memory, epoch)


def parse_topo_output(output, link_prefixes=("XGMI", "NV")):
    """GPU-link pairs from a topology matrix: rows start with "GPU<i>", a cell that starts with one
    of link_prefixes marks a direct link to the GPU of that column (rocm-smi --showtopotype prints
    XGMI / PCIE; nvidia-smi topo -m prints NV# -- legion_server.py:8-21 handles the latter)."""
    connections = []
    gpu_lines = [line for line in output.splitlines() if line.startswith("GPU")]
    for i, line in enumerate(gpu_lines):
        for j, elem in enumerate(line.split()[1:]):
            if any(elem.startswith(p) for p in link_prefixes):
                connections.append((i, j))
    return connections


def largest_clique_size(connections):
    """Size of the largest fully connected GPU group (Bron-Kerbosch); 1 when there are no links
    (legion_server.py:29-37 uses networkx.find_cliques for the same number)."""
    adj = {}
    for a, b in connections:
        if a == b:
            continue
        adj.setdefault(a, set()).add(b)
        adj.setdefault(b, set()).add(a)
    best = 1 if not adj else 0

    def expand(r, p, x):
        nonlocal best
        if not p and not x:
            best = max(best, len(r))
            return
        pivot = max(p | x, key=lambda v: len(adj[v] & p))
        for v in list(p - adj[pivot]):
            expand(r | {v}, p & adj[v], x & adj[v])
            p = p - {v}
            x = x | {v}

    if adj:
        expand(set(), set(adj), set())
    return max(best, 1)


def get_xgmi_topology():
    for cmd in (["rocm-smi", "--showtopotype"], ["/opt/rocm/bin/rocm-smi", "--showtopotype"]):
        try:
            res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=60)
            return parse_topo_output(res.stdout)
        except (OSError, subprocess.SubprocessError):
            continue
    return []


def cache_agg_mode_for(group_size):
    return math.log2(group_size)      # legion_server.py:106; the binary takes atoi of it


def server_binary():
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "sampling_server")


def Run(args):
    if args.dataset_name not in DATASETS:
        print("invalid dataset path")
        return 1
    with open("meta_config", "w") as file:
        file.write(meta_config_line(args.dataset_path, args.dataset_name, args.train_batch_size,
                                    args.cache_memory, args.epoch))
    gpu_number = args.gpu_number
    if args.usenvlink == 1:
        group_size = largest_clique_size(get_xgmi_topology())
        group_size = max(1, min(group_size, gpu_number))
        group_size = 1 << int(math.log2(group_size))
        print(f"xGMI clique size: {group_size}, Number of xGMI cliques: {int(gpu_number / group_size)}")
        cache_agg_mode = cache_agg_mode_for(group_size)
    else:
        cache_agg_mode = 0
    # The reference ignores the binary's status (legion_server.py:110).  Here it is returned as a process exit
    # code: subprocess.call gives the exit status itself (os.system's raw wait status, e.g. 256 for exit(1),
    # would wrap to 0 in sys.exit), and a death by signal n becomes 128 + n like a shell reports it.
    rc = subprocess.call([server_binary(), str(gpu_number), str(int(cache_agg_mode))] +
                         [str(f) for f in parse_fanout(args.fanout)])
    return 128 - rc if rc < 0 else rc


def parse_fanout(value):
    """--fanout accepts "25,10", "[25,10]" or the reference's argparse type=list spelling."""
    if isinstance(value, (list, tuple)) and all(isinstance(v, int) for v in value):
        return list(value)
    text = "".join(value) if isinstance(value, (list, tuple)) else str(value)
    return [int(t) for t in text.replace("[", " ").replace("]", " ").replace(",", " ").split()]


def build_argparser():
    argparser = argparse.ArgumentParser("Legion Server.")
    argparser.add_argument('--dataset_path', type=str, default="./dataset")
    argparser.add_argument('--dataset_name', type=str, default="ukunion")
    argparser.add_argument('--train_batch_size', type=int, default=8000)
    argparser.add_argument('--fanout', type=str, default="[25,10]")
    argparser.add_argument('--gpu_number', type=int, default=2)
    argparser.add_argument('--epoch', type=int, default=2)
    argparser.add_argument('--cache_memory', type=int, default=38000000)
    argparser.add_argument('--usenvlink', type=int, default=1)
    return argparser


def main(argv=None):
    return Run(build_argparser().parse_args(argv))
