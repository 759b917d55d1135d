"""Builds the in-tree native artefacts for gfx950 with hipcc.

    python -m legion_amd.build            # liblegion_hip.so + bin/sampling_server + the `ipc_service` torch extension, each when its
                                          # sources are newer (--no-trainer: skip the extension; --force: everything)

hipcc cross-compiles without a GPU.  Outputs stay inside the package directory (git-ignored,
shipped to the GPU box by gpurun).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "liblegion_hip.so")
BIN = os.path.join(HERE, "bin", "sampling_server")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

SOURCES = ["kernels_sample.hip", "kernels_gather.hip", "kernels_cache.hip", "kernels_synth.hip",
           "storage.hip", "tuning.hip", "link_counters.hip", "collective.hip", "markers.hip", "cache.hip", "operators.hip", "pipeline.hip", "ipc_env.hip", "server.hip"]
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-fno-fast-math", "-Wall",
         "-Wno-unused-result", "-Wno-unused-function"] + os.environ.get("LEGION_EXTRA_HIPCC_FLAGS", "").split()


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force=False, verbose=True):
    objdir = os.path.join(HERE, "_obj")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, "legion_core.h"), os.path.join(CSRC, "runner_schedule.h"), os.path.join(HERE, "..", "include", "legion_hip.h")]
    objs, procs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src + ".o")
        objs.append(o)
        if force or _newer(o, [s] + headers):
            cmd = [HIPCC] + FLAGS + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
    failed = [src for src, p in procs if p.wait() != 0]
    if failed:
        raise RuntimeError(f"hipcc failed for {failed}")
    if force or procs or not os.path.exists(LIB):
        cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs + ["-lpthread", "-lrt", "-ldl", "-L/opt/rocm/lib", "-lrccl",
                                                                                        "-Wl,-rpath,/opt/rocm/lib"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    main_src = os.path.join(CSRC, "main.cpp")
    if force or _newer(BIN, [main_src, LIB]):
        os.makedirs(os.path.dirname(BIN), exist_ok=True)
        cmd = [HIPCC, "-O2", "-std=c++17", main_src, "-o", BIN, f"-L{HERE}", "-llegion_hip",
               "-Wl,-rpath,$ORIGIN/.."]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    # the stand-alone probe of the file-descriptor convention of /opt/rocm's HIP runtime (tests/test_gpu_boundary.py)
    probe_src = os.path.join(HERE, "..", "tools", "micro", "vmm_convention_probe.cpp")
    probe_bin = os.path.join(HERE, "bin", "vmm_convention_probe")
    if force or _newer(probe_bin, [probe_src, os.path.join(HERE, "trainer", "vmm_probe.h")]):
        cmd = [HIPCC, "-O2", "-std=c++17", probe_src, "-o", probe_bin]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


def build_trainer(verbose=True, force=False):
    tdir = os.path.join(HERE, "trainer")
    import glob
    built = glob.glob(os.path.join(tdir, "ipc_service*.so"))
    srcs = [os.path.join(tdir, "ipc_service.cpp"), os.path.join(tdir, "vmm_probe.h"), os.path.join(tdir, "setup.py")]
    if built and not force and not _newer(built[0], srcs):
        return built[0]
    env = dict(os.environ, PYTORCH_ROCM_ARCH=ARCH)
    # --force: setuptools compares only the .cpp with its object; a changed header (vmm_probe.h) must rebuild too
    cmd = [sys.executable, "setup.py", "build_ext", "--inplace", "--force"]
    if verbose:
        print("(cd trainer &&", " ".join(cmd) + ")", flush=True)
    subprocess.check_call(cmd, cwd=tdir, env=env)


def build_variant(name, extra_flags, verbose=True):
    """liblegion_hip.so with other compile-time constants, for tuning sweeps:
        python -m legion_amd.build --variant v16 -DLG_LDS_BUCKET_BITS=4 ...
    -> tools/lds_tuning/variants/<name>/liblegion_hip.so (git-ignored; travels to the GPU box).  Select it at run time with
    LEGION_HIP_LIB=<path> (legion_amd/lib.py); the library in place is never overwritten."""
    out = os.path.join(HERE, "..", "tools", "lds_tuning", "variants", name)
    os.makedirs(out, exist_ok=True)
    objs, procs = [], []
    # LEGION_VARIANT_ONLY="kernels_sample.hip ...": the constants only reach these sources; the rest is the build in place
    only = os.environ.get("LEGION_VARIANT_ONLY", "").split()
    if only:
        build_lib(verbose=False)
    keep = []
    for src in SOURCES:
        if only and src not in only:
            keep.append(os.path.join(HERE, "_obj", src + ".o"))
            continue
        o = os.path.join(out, src + ".o")
        objs.append(o)
        cmd = [HIPCC] + FLAGS + ["-w"] + list(extra_flags) + ["-c", os.path.join(CSRC, src), "-o", o]
        procs.append((src, subprocess.Popen(cmd)))
    failed = [src for src, p in procs if p.wait() != 0]
    if failed:
        raise RuntimeError(f"hipcc failed for {failed}")
    lib = os.path.join(out, "liblegion_hip.so")
    subprocess.check_call([HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", lib] + objs + keep +
                          ["-lpthread", "-lrt", "-ldl", "-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib"])
    for o in objs:
        os.unlink(o)
    if verbose:
        print(lib)
    return lib


if __name__ == "__main__":
    if "--variant" in sys.argv:
        i = sys.argv.index("--variant")
        build_variant(sys.argv[i + 1], sys.argv[i + 2:])
        sys.exit(0)
    build_lib(force="--force" in sys.argv)
    if "--no-trainer" not in sys.argv:      # (by its time stamps, like the library: a stale trainer end once rode along for half a day of GPU runs)
        build_trainer(force="--force" in sys.argv)
