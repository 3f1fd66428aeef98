"""Build csrc/*.hip into csrc/libcddpm_hip.so for gfx950 with hipcc (in-tree, no JIT cache).

hipcc cross-compiles without a GPU, so this runs in the build container; the .so travels to the GPU box.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
SOURCES = ["conv_mfma.hip", "conv_x6.hip", "norm_kernels.hip", "small_kernels.hip", "attention.hip", "simplex.hip", "encoder.hip", "eval_post.hip", "train_kernels.hip", "encoder_train.hip", "cddpm_api.hip"]
LIB = os.path.join(CSRC, "libcddpm_hip.so")
OBJ = os.path.join(CSRC, "_obj")      # object files: git-ignored and .gpurunignore-d (only the .so travels to the GPU box)
ARCH = "gfx950"


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found (need ROCm); cannot build libcddpm_hip.so")
    return exe


def lib_is_current() -> bool:
    if not os.path.exists(LIB):
        return False
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + ["kernels.h", "conv_split.h"]]
    deps.append(os.path.join(os.path.dirname(os.path.dirname(CSRC)), "include", "cddpm.h"))
    return all(os.path.getmtime(d) <= t for d in deps if os.path.exists(d))


def build_lib(force: bool = False, verbose: bool = False, defines=(), tag: str = "", conv_src: str = "") -> str:
    """defines/tag: experimental A/B builds, e.g. defines=["CDDPM_STAMPS"], tag="stamps" -> libcddpm_hip_stamps.so
    (loaded through the CDDPM_LIB environment variable by tools/conv_ab.py); the product build has neither."""
    lib_out = LIB if not tag else LIB.replace(".so", f"_{tag}.so")
    if not force and not tag and lib_is_current():
        return LIB
    hipcc = _hipcc()
    os.makedirs(OBJ, exist_ok=True)
    flags = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"] + [f"-D{d}" for d in defines]
    objs = []

    def compile_one(src):
        obj = os.path.join(OBJ, src.replace(".hip", f"{('_' + tag) if tag else ''}.o"))
        # A/B of older conv kernels: an alternative file replaces conv_x6.hip if its name starts with conv_x6, else conv_mfma.hip
        swap = "conv_x6.hip" if os.path.basename(conv_src).startswith("conv_x6") else "conv_mfma.hip"
        path = conv_src if (conv_src and src == swap) else os.path.join(CSRC, src)
        deps = [path, os.path.join(CSRC, "kernels.h"), os.path.join(CSRC, "conv_split.h"),
                os.path.join(os.path.dirname(os.path.dirname(CSRC)), "include", "cddpm.h"), os.path.abspath(__file__)]
        stamp = obj + ".flags"          # an object is reused when it is newer than its inputs and was built with the same flags
        if (not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == " ".join(flags + [path])
                and all(os.path.getmtime(d) <= os.path.getmtime(obj) for d in deps if os.path.exists(d))):
            return obj
        cmd = [hipcc, *flags, f"-I{CSRC}", "-c", path, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        with open(stamp, "w") as f:
            f.write(" ".join(flags + [path]))
        return obj

    with ThreadPoolExecutor(max_workers=min(7, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    r = subprocess.run([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", lib_out, *objs], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc link failed:\n{r.stderr}")
    return lib_out


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
