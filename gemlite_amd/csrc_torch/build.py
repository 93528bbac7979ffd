#!/usr/bin/env python3
"""Build gemlite_amd/_fast.so (fast_forward.cpp) in-tree: the C++ eager host path.  Plain g++ against the installed torch headers
and libraries and libgemlite_hip.so (rpath-relative), no JIT cache: the .so travels with the repository snapshot.
    python gemlite_amd/csrc_torch/build.py [--force]"""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
SRC = os.path.join(HERE, "fast_forward.cpp")
OUT = os.path.join(PKG, "_fast.so")


def main(force=False):
    deps = [SRC, os.path.join(PKG, "..", "include", "gemlite_hip.h"), os.path.abspath(__file__)]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps):
        return OUT
    import torch
    from torch.utils import cpp_extension as ce
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = ["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-Wall", "-Wno-deprecated-declarations", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch.compiled_with_cxx11_abi())}",
           *[f"-I{p}" for p in ce.include_paths()], f"-I{rocm}/include", f"-I{sysconfig.get_paths()['include']}",
           SRC, "-o", OUT, f"-L{tlib}", "-ltorch_python", "-ltorch", "-ltorch_cpu", "-ltorch_hip", "-lc10", "-lc10_hip",
           f"-L{os.path.join(PKG, 'csrc')}", "-lgemlite_hip", "-Wl,-rpath,$ORIGIN/csrc", f"-Wl,-rpath,{tlib}"]
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    print("built", main(force="--force" in sys.argv))
