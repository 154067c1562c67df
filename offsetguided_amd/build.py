"""Build libog_decoder.so (hand-written HIP kernels + C ABI) for gfx950 with hipcc.

In-tree build: the .so sits next to this file, is git-ignored, and travels to the GPU box with
the repository snapshot.  `python -m offsetguided_amd.build` or `build()`.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libog_decoder.so")
SOURCES = ["abi.cpp", "nms_topk.hip", "upsample.hip", "collect.hip", "group.hip", "flip.hip", "epilogue.hip", "losses.hip", "conv3x3.hip", "conv_band.hip", "encoder.hip", "stem.hip", "preprocess.hip"]
# the 16-bit-type specific files are compiled a second time for fp16 (csrc/lp_dtype.h)
F16_SOURCES = ["conv3x3.hip", "conv_band.hip", "epilogue.hip", "stem.hip"]
ARCH = "gfx950"
# -ffp-contract=off: every FMA in the kernels is explicit (bit-exact parity with torch-CPU fp32);
# correctly-rounded fp32 divide/sqrt is hipcc's default and must stay on (no -ffast-math).
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-ffp-contract=off",
         f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function"]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libog_decoder.so cannot be built")
    return exe


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(ROOT, "include", "og_decoder.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    inc = ["-I", os.path.join(ROOT, "include"), "-I", CSRC]
    objs, procs = [], []
    for src, extra, tag in [(s_, [], "") for s_ in SOURCES] + [(s_, ["-DOG_DT_F16=1"], "_f16") for s_ in F16_SOURCES]:
        obj = os.path.join(objdir, os.path.splitext(src)[0] + tag + ".o")
        cmd = [_hipcc(), *FLAGS, *extra, *inc, "-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src + tag, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out.decode()}")
        if verbose and out:
            print(out.decode())
    cmd = [_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB + ".tmp", *objs]
    subprocess.check_call(cmd)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
