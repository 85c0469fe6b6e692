"""In-tree build of the native libraries (hipcc cross-compiles gfx950 without a GPU)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.abspath(__file__))


def lib_dir():
    # TPC_LIB_DIR: load the native libraries from another directory (A/B comparisons of builds)
    return os.environ.get("TPC_LIB_DIR") or os.path.join(ROOT, "lib")


def build_all(verbose=False):
    """Compile every HIP kernel for gfx950 and the C++ host layer. Idempotent (make)."""
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(["make", "-s", "-j4", "-C", os.path.join(ROOT, "csrc")], stdout=out)
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host")], stdout=out)
    for name in ("libtwopaco_hip.so", "libtwopaco_host.so"):
        if not os.path.exists(os.path.join(lib_dir(), name)):
            raise RuntimeError("build did not produce " + name)
