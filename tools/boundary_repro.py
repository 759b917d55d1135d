"""Repro loop for a rare start-up hang of server <-> trainer: starts the sampling_server binary on a tiny data set and a
fake trainer process N times; on a stall prints which side died / is stuck (server exit code or signal, log tails)."""
import os, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from legion_amd import synth

def main():
    n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    tmp = tempfile.mkdtemp(prefix="legion_repro_", dir="/tmp")
    scale, D, B = 11, 24, 48
    indptr, col = synth.rmat_csr_numpy(scale, 8, 20231)
    N = indptr.size - 1
    ds = os.path.join(tmp, "ds") + "/"
    os.makedirs(ds)
    indptr.astype(np.int64).tofile(ds + "edge_src"); col.astype(np.int32).tofile(ds + "edge_dst")
    synth.features_numpy(0, N, D, 7).tofile(ds + "features"); (np.arange(N) % 47).astype(np.int32).tofile(ds + "labels")
    perm = np.random.RandomState(3).permutation(N).astype(np.int32)
    perm[:500].tofile(ds + "trainingset"); perm[500:590].tofile(ds + "validationset"); perm[590:640].tofile(ds + "testingset")
    bad = 0
    for it in range(n_iter):
        work = os.path.join(tmp, f"run{it}")
        os.makedirs(work)
        open(os.path.join(work, "meta_config"), "w").write(f"{ds} {B} {N} {col.size} {D} 500 90 50 60000 2")
        ns = f"_r{os.getpid()}_{it}"
        env = dict(os.environ, LEGION_IPC_NAMESPACE=ns)
        log = open(os.path.join(work, "server.log"), "w")
        server = subprocess.Popen([os.path.join(ROOT, "legion_amd", "bin", "sampling_server"), "1", "0", "5", "3"], cwd=work, env=env,
                                  stdout=log, stderr=subprocess.STDOUT, stdin=subprocess.DEVNULL)
        t0 = time.time()
        while "System is ready for serving" not in open(os.path.join(work, "server.log")).read():
            if server.poll() is not None or time.time() - t0 > 120:
                print(f"iter {it}: server not ready, rc={server.poll()}"); break
            time.sleep(0.05)
        tl = open(os.path.join(work, "trainer.log"), "w")
        trainer = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "fake_trainer.py"), "0", str(D), "2", os.path.join(work, "t.npz")],
                                   env=env, cwd=ROOT, stdout=tl, stderr=subprocess.STDOUT, stdin=subprocess.DEVNULL)
        try:
            trainer.wait(timeout=40)
            server.wait(timeout=20)
            ok = trainer.returncode == 0 and server.returncode == 0
        except subprocess.TimeoutExpired:
            ok = False
        if not ok:
            bad += 1
            print(f"iter {it}: STALL/FAIL trainer rc={trainer.poll()} server rc={server.poll()}")
            print("  server log tail:", open(os.path.join(work, "server.log")).read()[-300:].replace("\n", " | "))
            print("  trainer log tail:", open(os.path.join(work, "trainer.log")).read()[-300:].replace("\n", " | "))
            if server.poll() is None:
                subprocess.call(["bash", "-c", f"cat /proc/{server.pid}/status | grep -E 'State|Threads'; for t in /proc/{server.pid}/task/*; do echo $(cat $t/comm) $(cat $t/wchan 2>/dev/null) $(grep State $t/status); done"])
        for p in (trainer, server):
            if p.poll() is None:
                p.kill()
        log.close(); tl.close()
        for name in os.listdir("/dev/shm"):
            if name.endswith(ns):
                os.unlink(os.path.join("/dev/shm", name))
    print(f"{n_iter - bad} of {n_iter} runs clean")
    subprocess.call(["rm", "-rf", tmp])

if __name__ == "__main__":
    main()
