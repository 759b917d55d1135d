"""Starting the `sampling_server` binary from a test."""
import subprocess
import sys
import time


def start_server(argv, cwd, env, log_path, timeout=400, attempts=3):
    """Starts the server and waits for its readiness line; returns (process, open log file).

    On this pool (ROCm 7.2, dmabuf IPC) about one server start in a few hundred finds hipIpcGetMemHandle failing with
    'invalid argument' for every block of the process (pointer attributes fine, HSA_ENABLE_IPC_MODE_LEGACY=0 set) and exits
    before it is ready -- the next start is fine.  Whatever supervises a server restarts it; so does this helper, loudly.
    Any other early exit fails the test with the server's log."""
    for attempt in range(attempts):
        log = open(log_path, "w")
        proc = subprocess.Popen(argv, cwd=cwd, env=env, stdout=log, stderr=subprocess.STDOUT, stdin=subprocess.DEVNULL)
        deadline = time.time() + timeout          # a cold box pages the ROCm libraries in first
        while True:
            text = open(log_path).read()
            if "System is ready for serving" in text:
                return proc, log
            if proc.poll() is not None:
                log.close()
                if "hipIpcGetMemHandle" in text and attempt + 1 < attempts:
                    print(f"sampling_server could not export its IPC buffers (attempt {attempt + 1}); restarting it:\n{text[-600:]}",
                          file=sys.stderr)
                    break
                raise AssertionError(text)
            if time.time() > deadline:
                proc.kill()
                log.close()
                raise AssertionError("server did not become ready:\n" + text[-3000:])
            time.sleep(0.1)
    raise AssertionError("unreachable")
