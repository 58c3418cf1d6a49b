"""Build libfaststyle_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["fs_conv.hip", "fs_wino.hip", "fs_wgrad.hip", "fs_elem.hip", "fs_fold.hip", "fs_io.hip", "fs_tnet.hip", "fs_bf16.hip", "fs_vgg.hip", "fs_api.hip"]
OUT = os.path.join(HERE, "libfaststyle_hip.so")


def _newest(paths):
    return max(os.path.getmtime(p) for p in paths)


def build(force=False, verbose=False):
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    deps = srcs + [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")] + \
        [os.path.join(os.path.dirname(HERE), "include", "faststyle_hip.h")]
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= _newest(deps):
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-result",
           "-Wno-pass-failed", "-I", CSRC] + srcs + ["-o", OUT]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
