"""In-tree build of the HIP extension (librgbdfe.so) for gfx950."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(_HERE, "librgbdfe.so")


def build(verbose=False):
    """Compile every HIP source with hipcc --offload-arch=gfx950 (cross-compiles without a GPU)."""
    env = dict(os.environ)
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    env["HIPCC"] = hipcc
    out = subprocess.run(["make", "-C", CSRC, "-j4"], env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, text=True)
    if verbose or out.returncode != 0:
        print(out.stdout)
    if out.returncode != 0:
        raise RuntimeError("building librgbdfe.so failed")
    if not os.path.exists(LIB):
        raise RuntimeError("librgbdfe.so was not produced")
    return LIB


if __name__ == "__main__":
    print(build(verbose=True))
