"""Build recipe for libmpl_hip.so (gfx950 only, hipcc cross-compiles without a GPU).

    python -m openmpl_amd.build [--force]

The library is built IN-TREE (openmpl_amd/lib/libmpl_hip.so) so that it travels to the GPU
box with the repository snapshot; it is git-ignored.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libmpl_hip.so")
SOURCES = ["api.hip", "spt.hip", "ln_gemm.hip", "h2_gemm.hip", "h2n_gemm.hip", "h2d_gemm.hip", "b1_gemm.hip", "sm_stack.hip", "token_attention.hip", "fuse_head.hip", "heads.hip", "metrics.hip", "inputs.hip"]
HEADERS = [os.path.join(CSRC, "common.hpp"), os.path.join(CSRC, "gemm_common.hpp"), os.path.join(CSRC, "h2_phase.hpp"), os.path.join(CSRC, "fuse_head.hpp"), os.path.join(os.path.dirname(HERE), "include", "mpl_hip.h")]
ARCH = "gfx950"


class CompilerMissing(RuntimeError):
    """No hipcc on this machine (the only build failure cabi.load() may tolerate, and only with an existing library)."""


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise CompilerMissing("hipcc not found (ROCm toolchain required to build libmpl_hip.so)")


STAMP_PATH = LIB_PATH + ".srchash"


def source_hash() -> str:
    """Content hash of everything the library is built from (mtimes do not survive the snapshot copy to a GPU box)."""
    import hashlib
    h = hashlib.sha256()
    h.update(os.environ.get("MPL_HIPCC_FLAGS", "").encode())      # a build with other flags is another library
    for d in [os.path.join(CSRC, s) for s in SOURCES] + HEADERS:
        h.update(os.path.basename(d).encode())
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH) or not os.path.exists(STAMP_PATH):
        return True
    try:
        return open(STAMP_PATH).read().strip() != source_hash()
    except OSError:
        return True


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile + link under an exclusive file lock, into private temporaries, and publish the library with one
    atomic rename: N ranks starting together (torchrun / bench.py --gpus N) never see a half-written .so, and all but
    the first find it up to date once they get the lock."""
    if not force and not needs_build():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    import fcntl
    with open(os.path.join(LIB_DIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not needs_build():
                return LIB_PATH
            return _build_locked(verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(verbose: bool) -> str:
    cc = _hipcc()
    srchash = source_hash()          # of the sources as they are NOW: an edit during the build leaves the stamp stale
    objs = []
    procs = []
    tag = ".%d" % os.getpid()
    tmp_so = LIB_PATH + tag
    try:
        for src in SOURCES:
            obj = os.path.join(LIB_DIR, src.replace(".hip", tag + ".o"))
            cmd = [cc, "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC"] + os.environ.get("MPL_HIPCC_FLAGS", "").split() + \
                  ["-c", os.path.join(CSRC, src), "-o", obj]
            if verbose:
                print(" ".join(cmd))
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
            objs.append(obj)
        failed = None
        for src, p in procs:
            out, _ = p.communicate()
            if p.returncode != 0 and failed is None:
                failed = "hipcc failed on %s:\n%s" % (src, out.decode(errors="replace"))
        if failed:
            raise RuntimeError(failed)
        cmd = [cc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", tmp_so] + objs
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stdout.decode(errors="replace"))
        os.replace(tmp_so, LIB_PATH)
        with open(STAMP_PATH + tag, "w") as f:
            f.write(srchash + "\n")
        os.replace(STAMP_PATH + tag, STAMP_PATH)
        return LIB_PATH
    finally:
        # the PID-tagged temporaries never outlive the build, whether it succeeded, failed or was interrupted
        for _, p in procs:
            if p.poll() is None:
                p.kill()
                p.wait()
        for f in objs + [tmp_so, STAMP_PATH + tag]:
            try:
                os.remove(f)
            except OSError:
                pass


# ---------------------------------------------------------------------------------------------- the torch extension
# csrc/torch_ext.cpp: TORCH_LIBRARY operators (openmpl_amd::bind / lift / unbind) over the C ABI -- host C++ only (g++), links
# against torch, reaches libmpl_hip.so through entry-point addresses handed over at load time.  Built in-tree like the library.
EXT_SRC = os.path.join(CSRC, "torch_ext.cpp")
EXT_PATH = os.path.join(LIB_DIR, "mpl_torch_ext.so")
EXT_STAMP = EXT_PATH + ".srchash"


def ext_source_hash() -> str:
    import hashlib
    import torch
    h = hashlib.sha256()
    h.update(torch.__version__.encode())
    for d in (EXT_SRC, HEADERS[-1]):
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def ext_needs_build() -> bool:
    if not os.path.exists(EXT_PATH) or not os.path.exists(EXT_STAMP):
        return True
    try:
        return open(EXT_STAMP).read().strip() != ext_source_hash()
    except OSError:
        return True


def build_torch_ext(force: bool = False, verbose: bool = False) -> str:
    if not force and not ext_needs_build():
        return EXT_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    import fcntl
    import torch
    from torch.utils import cpp_extension as ce
    cxx = os.environ.get("CXX") or shutil.which("g++") or shutil.which("c++")
    if not cxx:
        raise CompilerMissing("no C++ compiler for the torch extension")
    with open(os.path.join(LIB_DIR, ".build_ext.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not ext_needs_build():
                return EXT_PATH
            srchash = ext_source_hash()
            tag = ".%d" % os.getpid()
            tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
            inc = [i for p in ce.include_paths(True) for i in ("-I", p)] + ["-I", "/opt/rocm/include"]
            cmd = [cxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
                   "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI)] + inc + \
                  [EXT_SRC, "-o", EXT_PATH + tag, "-L", tlib, "-lc10", "-ltorch_cpu", "-ltorch", "-lc10_hip", "-ltorch_hip", "-lamdhip64",
                   "-Wl,-rpath," + tlib]
            if verbose:
                print(" ".join(cmd))
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
            if r.returncode != 0:
                try:
                    os.remove(EXT_PATH + tag)
                except OSError:
                    pass
                raise RuntimeError("torch extension build failed:\n" + r.stdout.decode(errors="replace"))
            os.replace(EXT_PATH + tag, EXT_PATH)
            with open(EXT_STAMP + tag, "w") as f:
                f.write(srchash + "\n")
            os.replace(EXT_STAMP + tag, EXT_STAMP)
            return EXT_PATH
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_torch_ext(force="--force" in sys.argv, verbose=True))
