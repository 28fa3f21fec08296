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


def _stale(obj, src, headers):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(p) > t for p in [src, __file__] + headers)


def build(force=False, verbose=True):
    """Incremental: a source is recompiled when it, this recipe or any header is newer than its object (headers are
    shared by nearly every translation unit); at most ``PGV_BUILD_JOBS`` (default 8) hipcc processes at a time."""
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    headers = glob.glob(os.path.join(SRC, "*.h")) + [os.path.join(HERE, "..", "include", "pgv_hip.h")]
    objs = []
    todo = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    for s in sources():
        o = os.path.join(HERE, "build", os.path.basename(s) + ".o")
        objs.append(o)
        if force or _stale(o, s, headers):
            cmd = [hipcc] + [f for f in FLAGS if f != "-shared"] + EXTRA.get(os.path.basename(s), []) + ["-c", s, "-o", o]
            todo.append((s, cmd))
    # the long translation units first
    todo.sort(key=lambda sc: -os.path.getsize(sc[0]))
    jobs = max(1, int(os.environ.get("PGV_BUILD_JOBS", "8")))
    running = []
    while todo or running:
        while todo and len(running) < jobs:
            s, cmd = todo.pop(0)
            running.append((s, subprocess.Popen(cmd)))
        s, p = running.pop(0)
        if p.wait() != 0:
            for _, q in running:
                q.kill()
            raise RuntimeError(f"hipcc failed on {s}")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
