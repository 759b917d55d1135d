"""A GraphSAGE trainer against `ipc_service`, shaped like training_backend/legion_graphsage.py (one process per GPU:
initialize -> get_steps -> per step get_next / get_block_size / model / synchronize -> finalize) but with the two DGL
pieces -- `DGLBlock` + `dgl.nn.SAGEConv(aggregator 'mean')` -- written in plain PyTorch, because DGL is not part of this
image.  A block is used exactly as `create_dgl_block(src, dst, n_src, n_dst)` defines it (legion_graphsage.py:66-70):
an edge e carries h[src[e]] (a row of the n_src input rows) to output row dst[e] (< n_dst), and the output rows are the
first n_dst input rows (the batch's node order is cumulative: seeds, hop 1, hop 2, ...).

    python examples/graphsage_torch.py --device 0 --features_num 16 --hidden_dim 32 --class_num 2 --hops_num 2 \
        --epoch 8 [--drop_rate 0] [--lr 0.01] [--report out.json]

Start `sampling_server` (or legion_server.py) first; run one trainer per server GPU."""
import argparse
import json
import os
import sys

import torch
import torch.nn as nn
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "legion_amd", "trainer"))


class SAGEConvMean(nn.Module):
    """dgl.nn.SAGEConv(in, out, 'mean'): fc_self(h_dst) + fc_neigh(mean of the in-neighbours' h_src), bias."""

    def __init__(self, in_feats, out_feats):
        super().__init__()
        self.fc_self = nn.Linear(in_feats, out_feats, bias=False)
        self.fc_neigh = nn.Linear(in_feats, out_feats, bias=False)
        self.bias = nn.Parameter(torch.zeros(out_feats))

    def forward(self, block, h):
        src, dst, n_src, n_dst = block
        assert h.shape[0] == n_src, (h.shape, n_src)
        acc = torch.zeros(n_dst, h.shape[1], device=h.device, dtype=h.dtype)
        acc.index_add_(0, dst.long(), h[src.long()])
        deg = torch.zeros(n_dst, device=h.device, dtype=h.dtype)
        deg.index_add_(0, dst.long(), torch.ones_like(dst, dtype=h.dtype))
        mean = acc / deg.clamp(min=1).unsqueeze(1)
        return self.fc_self(h[:n_dst]) + self.fc_neigh(mean) + self.bias


class SAGE(nn.Module):                       # legion_graphsage.py:36-64
    def __init__(self, in_feats, n_hidden, n_classes, n_layers, dropout):
        super().__init__()
        dims = [in_feats] + [n_hidden] * (n_layers - 1) + [n_classes]
        self.layers = nn.ModuleList(SAGEConvMean(dims[i], dims[i + 1]) for i in range(n_layers))
        self.dropout = nn.Dropout(dropout)

    def forward(self, blocks, x):
        h = x
        for l, (layer, block) in enumerate(zip(self.layers, blocks)):
            h = layer(block, h)
            if l != len(self.layers) - 1:
                h = self.dropout(F.relu(h))
        return h


def next_batch(ipc_service, feat_len, hops):
    out = ipc_service.get_next(feat_len)               # [ids, feats, labels, (src, dst) for h = H .. 1]
    sizes = ipc_service.get_block_size()               # [n_src, n_dst for h = H .. 1]
    feats, labels = out[1], out[2]
    blocks = [(out[3 + 2 * k], out[4 + 2 * k], sizes[2 * k], sizes[2 * k + 1]) for k in range(hops)]
    return feats, labels, blocks


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--features_num", type=int, default=128)
    ap.add_argument("--hidden_dim", type=int, default=256)
    ap.add_argument("--class_num", type=int, default=47)
    ap.add_argument("--hops_num", type=int, default=2)
    ap.add_argument("--drop_rate", type=float, default=0.5)
    ap.add_argument("--lr", type=float, default=0.003)
    ap.add_argument("--epoch", type=int, default=10)
    ap.add_argument("--report", type=str, default="")
    a = ap.parse_args()

    import ipc_service
    torch.cuda.set_device(a.device % torch.cuda.device_count())
    dev = torch.device("cuda", torch.cuda.current_device())
    torch.manual_seed(0)
    ipc_service.initialize()
    train_steps, valid_steps, test_steps = ipc_service.get_steps()
    model = SAGE(a.features_num, a.hidden_dim, a.class_num, a.hops_num, a.drop_rate).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=a.lr)
    history = []

    def evaluate(steps):
        model.eval()
        hit = tot = 0
        with torch.no_grad():
            for _ in range(steps):
                feats, labels, blocks = next_batch(ipc_service, a.features_num, a.hops_num)
                pred = model(blocks, feats).argmax(dim=1)
                hit += int((pred == labels.long()).sum())
                tot += int(labels.numel())
                del feats, labels, blocks
                torch.cuda.synchronize()
                ipc_service.synchronize()
        return hit / max(tot, 1)

    for ep in range(a.epoch):
        model.train()
        loss_sum = 0.0
        for _ in range(train_steps):
            feats, labels, blocks = next_batch(ipc_service, a.features_num, a.hops_num)
            loss = F.cross_entropy(model(blocks, feats), labels.long())
            opt.zero_grad()
            loss.backward()
            opt.step()
            loss_sum += float(loss)
            del feats, labels, blocks
            torch.cuda.synchronize()
            ipc_service.synchronize()
        acc = evaluate(valid_steps)
        history.append({"epoch": ep, "train_loss": loss_sum / max(train_steps, 1), "valid_acc": acc})
        print(f"epoch {ep}: loss {history[-1]['train_loss']:.4f}  valid acc {acc:.4f}", flush=True)
    test_acc = evaluate(test_steps)
    print(f"test acc {test_acc:.4f}", flush=True)
    ipc_service.finalize()
    if a.report:
        json.dump({"history": history, "test_acc": test_acc, "steps": [train_steps, valid_steps, test_steps]}, open(a.report, "w"))


if __name__ == "__main__":
    main()
