"""A stand-in for training_backend/legion_graphsage.py's data path (lines 72-91, 121-128): walks the `ipc_service`
protocol for ONE server GPU -- initialize, then per step get_next -> get_block_size -> (the model would run here)
-> synchronize -- and dumps everything it was handed to an .npz for the test to compare with the oracle.
    python tests/fake_trainer.py <logical server gpu> <feature dim> <epochs> <out.npz>"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "legion_amd", "trainer"))


def main():
    dev, dim, epoch, out_path = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    os.environ["LEGION_IPC_DEVICE"] = str(dev)          # logical server GPU; this box may have fewer physical ones
    import torch
    import ipc_service
    torch.cuda.set_device(dev % torch.cuda.device_count())
    ipc_service.initialize()
    train, valid, test = ipc_service.get_steps()
    out = {"steps": np.array([train, valid, test], dtype=np.int32)}
    total = (train + valid) * epoch + test
    for i in range(total):
        t = ipc_service.get_next(dim)
        sizes = ipc_service.get_block_size()
        # the Python ABI of TB/ipc_service.cpp:44-79: int32 ids / labels / edge positions, float32 rows, all on the GPU
        assert t[0].dtype == torch.int32 and t[1].dtype == torch.float32 and t[2].dtype == torch.int32 and all(x.is_cuda for x in t)
        assert all(x.dtype == torch.int32 for x in t[3:]) and t[1].dim() == 2 and t[1].shape[1] == dim and t[1].shape[0] == t[0].shape[0]
        out[f"b{i}_ntensors"] = np.int32(len(t))
        out[f"b{i}_ids"] = t[0].cpu().numpy()
        out[f"b{i}_feats"] = t[1].cpu().numpy().view(np.uint32)
        out[f"b{i}_labels"] = t[2].cpu().numpy()
        for k in range((len(t) - 3) // 2):
            out[f"b{i}_src{k}"] = t[3 + 2 * k].cpu().numpy()
            out[f"b{i}_dst{k}"] = t[4 + 2 * k].cpu().numpy()
        out[f"b{i}_sizes"] = np.array(sizes, dtype=np.int32)
        del t
        torch.cuda.synchronize()
        ipc_service.synchronize()
    ipc_service.finalize()
    np.savez(out_path, **out)


if __name__ == "__main__":
    main()
