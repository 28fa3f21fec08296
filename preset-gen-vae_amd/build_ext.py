"""Build recipe of the C-ABI library: ``hipcc --offload-arch=gfx950 -shared`` over ``csrc/*.hip`` -> in-tree
``libpgv_hip.so`` (travels to the GPU box with the snapshot; git-ignored)."""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libpgv_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-result"]
# per-file extras: the FFT butterflies are complex arithmetic on register pairs - packing them into v_pk_* instructions
# (SLP vectoriser) costs more register moves than it saves: 1560 -> 1376 instructions per frame pair without it
EXTRA = {"stft_mel.hip": ["-fno-slp-vectorize"]}


def sources():
    return sorted(glob.glob(os.path.join(SRC, "*.hip")))


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = sources() + glob.glob(os.path.join(SRC, "*.h")) + [os.path.join(HERE, "..", "include", "pgv_hip.h")]
    return any(os.path.getmtime(p) > t for p in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    for s in sources():
        o = os.path.join(HERE, "build", os.path.basename(s) + ".o")
        objs.append(o)
        cmd = [hipcc] + [f for f in FLAGS if f != "-shared"] + EXTRA.get(os.path.basename(s), []) + ["-c", s, "-o", o]
        procs.append((s, subprocess.Popen(cmd)))
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {s}")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
