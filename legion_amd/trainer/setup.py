"""Builds the `ipc_service` PyTorch extension in-tree (training_backend/setup.py in the reference):
    PYTORCH_ROCM_ARCH=gfx950 python setup.py build_ext --inplace
The module is host code only (HIP runtime API + torch::from_blob), so it is a CppExtension linked
against libamdhip64; nothing is hipified."""
import os

from setuptools import setup
from torch.utils.cpp_extension import BuildExtension, CppExtension

ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
setup(
    name="ipcservice",
    ext_modules=[
        CppExtension("ipc_service", ["ipc_service.cpp"],
                     include_dirs=[os.path.join(ROCM, "include")],
                     library_dirs=[os.path.join(ROCM, "lib")],
                     libraries=["amdhip64", "rt", "pthread"],
                     define_macros=[("__HIP_PLATFORM_AMD__", "1")],
                     extra_compile_args=["-O2", "-std=c++17"])
    ],
    cmdclass={"build_ext": BuildExtension.with_options(use_ninja=False)})
