"""Builds libkrs_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m keras_rs_amd.build [--force]

One object per .hip under csrc/, linked into keras_rs_amd/libkrs_hip.so.  The
.so is git-ignored but travels to the GPU box with the repo snapshot.
"""

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libkrs_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-I", os.path.join(os.path.dirname(HERE), "include")]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "krs.h"))
    jobs = []
    objs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        # (embed_bag_bwd.hip: ~590 kernel instantiations -- four objects compiled side by side, see KRS_BWD_PART there)
        parts = [(f"_p{i}", [f"-DKRS_BWD_PART={i}"]) for i in range(4)] if s == "embed_bag_bwd.hip" else [("", [])]
        for suffix, defs in parts:
            obj = os.path.join(OBJ, s[:-4] + suffix + ".o")
            objs.append(obj)
            if force or _newer(obj, [src] + hdrs):
                jobs.append([HIPCC] + FLAGS + defs + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        return r

    for stale in set(os.path.join(OBJ, f) for f in os.listdir(OBJ) if f.endswith(".o")) - set(objs):
        os.remove(stale)          # (objects of sources that no longer exist)
    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if jobs or force or _newer(LIB, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
