"""ONE trainer process that outlives several server lives: per life it starts the `sampling_server` binary,
walks initialize -> [get_next -> get_block_size -> synchronize] x the whole schedule -> finalize, waits for the server to exit and
checks that the GPU's free memory is back where it was before the first life -- i.e. that finalize() unmapped the server's lane
arena (round 4 kept it mapped "until the process ends", which pinned the dead server's whole arena in HBM) -- and dumps what it
was handed to <work>/life<k>.npz for the test to compare with the oracle.
    python tests/long_lived_trainer.py <lives> <feature dim> <epochs> <work dir with meta_config> <fan-out ...>"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "legion_amd", "trainer"))
from tests.server_proc import start_server  # noqa: E402


def vram_held_mib():
    """What THIS process holds in VRAM according to the driver (amdgpu fdinfo of its drm / kfd descriptors, one entry per drm
    client).  hipMemGetInfo answers from the runtime's own book-keeping, which does not follow another process's chunks being
    mapped and unmapped here (tools/micro/vmm_release_probe.cpp)."""
    seen, total = set(), 0
    for name in os.listdir("/proc/self/fdinfo"):
        try:
            text = open(f"/proc/self/fdinfo/{name}").read()
        except OSError:
            continue
        cid, vram = None, None
        for ln in text.splitlines():
            if ln.startswith("drm-client-id:"):
                cid = ln.split()[1]
            elif ln.startswith("drm-memory-vram:"):
                vram = int(ln.split()[1])
        if vram is not None and cid not in seen:
            seen.add(cid)
            total += vram >> 10
    return total


def main():
    import hashlib
    lives, dim, epoch, work = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    fanout = sys.argv[5:]
    import torch
    import ipc_service
    torch.cuda.set_device(0)
    torch.zeros(1, device="cuda:0")
    torch.cuda.synchronize()
    # (no ipc_service.vmm_fd_convention() here: that explicit probe imports a chunk this process exported itself; the product path learns
    # the convention from the first descriptor the server sends, and this process should do nothing a trainer would not)
    dig = lambda t: int.from_bytes(hashlib.blake2b(t.contiguous().cpu().numpy().tobytes(), digest_size=8).digest(), "little")
    level = None                                # free memory after the first life (everything of this process is warm by then)
    for life in range(lives):
        server, log = start_server([os.path.join(ROOT, "legion_amd", "bin", "sampling_server"), "1", "0"] + fanout, work, dict(os.environ),
                                   os.path.join(work, f"server{life}.log"))
        try:
            ipc_service.initialize()
            train, valid, test = ipc_service.get_steps()
            total = (train + valid) * epoch + test
            digests, sizes = [], []
            for i in range(total):
                t = ipc_service.get_next(dim)
                sizes.append(list(ipc_service.get_block_size()))
                assert t[1].dim() == 2 and t[1].shape[1] == dim and t[1].shape[0] == t[0].shape[0]
                digests.append([dig(t[0]), dig(t[1].view(torch.int32)), dig(t[2])] + [dig(x) for x in t[3:]])
                del t
                torch.cuda.synchronize()
                ipc_service.synchronize()
            mid_free, mid_held = torch.cuda.mem_get_info(0)[0], vram_held_mib()
            ipc_service.finalize()
            server.wait(timeout=120)
            assert server.returncode == 0, open(os.path.join(work, f"server{life}.log")).read()[-2000:]
        finally:
            if server.poll() is None:
                server.kill()
            log.close()
        np.savez(os.path.join(work, f"life{life}.npz"), steps=np.array([train, valid, test], dtype=np.int32),
                 digests=np.array(digests, dtype=np.uint64), sizes=np.array(sizes, dtype=np.int32))
        torch.cuda.empty_cache()
        free, held = torch.cuda.mem_get_info(0)[0], vram_held_mib()
        print(f"life {life}: {total} batches; VRAM held by this process (driver's fdinfo) while attached {mid_held} MiB, after finalize + server "
              f"exit {held} MiB; hipMemGetInfo free {mid_free >> 20} -> {free >> 20} MiB", flush=True)
        if level is None:
            level = held
        # the dead server's arena (hundreds of MB per life) must not survive in this process's mappings
        assert mid_held >= held + 200, f"life {life}: the lane arena does not show in what this process holds ({mid_held} vs {held} MiB): nothing is measured"
        assert held <= level + 16, f"life {life}: this process still holds {held - level} MiB more VRAM than after the first life"
    print("all lives clean", flush=True)


if __name__ == "__main__":
    main()
