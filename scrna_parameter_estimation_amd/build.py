"""Build libmemento_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

The .so lands next to the sources (scrna_parameter_estimation_amd/csrc/libmemento_hip.so); it is
git-ignored but travels to the GPU box with the repo snapshot.
"""

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libmemento_hip.so")
# (source, extra flags).  hist/boot replay numpy's IEEE arithmetic: contraction must stay off there.
SOURCES = [
    ("runtime.hip", []),
    ("ingest.hip", []),
    ("moments.hip", []),
    ("hist.hip", ["-ffp-contract=off"]),
    ("boot.hip", ["-ffp-contract=off"]),
    ("contract.hip", ["-ffp-contract=off"]),   # mirrors numpy's unfused weighted sums (degenerate resampled columns)
    ("pairs.hip", []),
    ("simulate.hip", []),
]
COMMON = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]
# experiment hook: extra -D flags for kernel tuning (e.g. MM_EXTRA_DEFS="-DK1_UNROLL=8 -DMM_ITEM_ROWS=32")
COMMON += os.environ.get("MM_EXTRA_DEFS", "").split()


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(os.path.dirname(CSRC)), "include", "memento_hip.h"))
    headers.append(os.path.abspath(__file__))  # flag changes in this file rebuild everything
    objs, jobs = [], []
    for src, extra in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + headers):
            jobs.append([hipcc, *COMMON, *extra, "-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        if verbose and r.stderr.strip():
            print(r.stderr)

    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if jobs or force or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
