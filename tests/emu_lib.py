"""TEST INFRASTRUCTURE: builds the unmodified kernel sources against the fiber emulator in
tests/emu (host clang++, no GPU) and binds the resulting library with the SAME ctypes
prototypes and the SAME Engine code as the product, backed by numpy host arrays."""
import ctypes
import os
import subprocess

import numpy as np

from faststyle_amd import _lib, build as fsbuild, engine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_DIR = os.path.join(ROOT, "tests", "emu")
EMU_SO = os.path.join(EMU_DIR, "libfaststyle_emu.so")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


ASAN_RT = "/opt/rocm/lib/llvm/lib/clang/22/lib/linux/libclang_rt.asan-x86_64.so"


def build_emu_sanitized():
    """The same sources with -fsanitize=address,undefined (tests/test_fuzz_emu.py runs tools/fuzz_emu.py against it in a
    subprocess that preloads the ASan runtime): an out-of-range LDS or global index of a ragged-edge path is then an error
    even where it does not change a compared value.  Device buffers are numpy allocations (ASan intercepts malloc once its
    runtime is preloaded), a workgroup's LDS is a heap block of exactly the launch's size."""
    so = os.path.join(EMU_DIR, "asan", "libfaststyle_emu_asan.so")
    srcs = [os.path.join(fsbuild.CSRC, s) for s in fsbuild.SOURCES]
    deps = srcs + [os.path.join(fsbuild.CSRC, h) for h in os.listdir(fsbuild.CSRC) if h.endswith(".h")] + \
        [os.path.join(EMU_DIR, "hip", "hip_runtime.h"), os.path.join(ROOT, "include", "faststyle_hip.h"),
         os.path.join(ROOT, "include", "faststyle_io.h")]
    if os.path.exists(so) and os.path.getmtime(so) >= max(os.path.getmtime(p) for p in deps):
        return so
    os.makedirs(os.path.dirname(so), exist_ok=True)
    objs, procs = [], []
    for s in srcs:
        o = os.path.join(EMU_DIR, "asan", os.path.basename(s) + ".o")
        objs.append(o)
        procs.append(subprocess.Popen([CLANG, "-x", "c++", "-std=c++17", "-O1", "-g", "-fPIC", "-fsanitize=address,undefined",
                                       "-fno-sanitize=float-divide-by-zero,float-cast-overflow", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
                                       "-shared-libsan", "-Wno-psabi", "-Wno-pass-failed", "-I", EMU_DIR, "-I", fsbuild.CSRC, "-c", s, "-o", o]))
    for p in procs:
        if p.wait():
            raise RuntimeError("sanitized emulator build failed")
    subprocess.check_call([CLANG, "-shared", "-fsanitize=address,undefined", "-shared-libsan", "-o", so] + objs + ["-lpthread"])
    return so


def build_emu():
    if os.environ.get("FS_EMU_SANITIZE") == "1":
        return build_emu_sanitized()
    srcs = [os.path.join(fsbuild.CSRC, s) for s in fsbuild.SOURCES]
    deps = srcs + [os.path.join(fsbuild.CSRC, h) for h in os.listdir(fsbuild.CSRC) if h.endswith(".h")] + \
        [os.path.join(EMU_DIR, "hip", "hip_runtime.h"), os.path.join(ROOT, "include", "faststyle_hip.h"),
         os.path.join(ROOT, "include", "faststyle_io.h")]
    if os.path.exists(EMU_SO) and os.path.getmtime(EMU_SO) >= max(os.path.getmtime(p) for p in deps):
        return EMU_SO
    objs = []
    procs = []
    for s in srcs:
        o = os.path.join(EMU_DIR, os.path.basename(s) + ".o")
        objs.append(o)
        procs.append(subprocess.Popen([CLANG, "-x", "c++", "-std=c++17", "-O2", "-fPIC", "-Wno-psabi", "-Wno-pass-failed",
                                       "-I", EMU_DIR, "-I", fsbuild.CSRC, "-c", s, "-o", o]))
    for p in procs:
        if p.wait():
            raise RuntimeError("emulator build failed")
    subprocess.check_call([CLANG, "-shared", "-o", EMU_SO] + objs + ["-lpthread"])
    return EMU_SO


class NumpyMem(object):
    def empty(self, shape):
        return np.full(tuple(int(s) for s in shape), np.nan, dtype=np.float32)   # NaN-poisoned

    def zeros(self, shape):
        return np.zeros(tuple(int(s) for s in shape), dtype=np.float32)

    def from_numpy(self, a):
        return np.ascontiguousarray(a, dtype=np.float32).copy()

    def to_numpy(self, t):
        return t

    def ptr(self, t):
        if t is None:
            return None
        assert t.flags.c_contiguous and t.dtype == np.float32
        return t.ctypes.data

    def stream(self):
        return 0

    def upload_u8(self, a):
        return np.ascontiguousarray(a, dtype=np.uint8).copy()

    def ptr_u8(self, t):
        assert t.flags.c_contiguous and t.dtype == np.uint8
        return t.ctypes.data

    def gather_rows(self, store, idx):
        return store[np.asarray(idx, dtype=np.int64)].copy()

    def copy_row(self, store, src, dst):
        store[dst] = store[src]

    def device_index(self):
        return 0

    def view(self, t, offset, shape):
        n = int(np.prod(shape))
        return t.reshape(-1)[offset:offset + n].reshape(shape)


_engine = None


def emu_engine():
    global _engine
    if _engine is None:
        lib = _lib.bind(ctypes.CDLL(build_emu()))
        _engine = engine.Engine(mem=NumpyMem(), lib=lib)
    return _engine
