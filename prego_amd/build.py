"""In-tree build of libprego_amd.so for gfx950 (hipcc cross-compiles without a GPU).

    python -m prego_amd.build            # incremental
    python -m prego_amd.build --force
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libprego_amd.so")
ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-fPIC", "-std=c++17", f"--offload-arch={ARCH}", "-Wno-unused-result", "-Wno-inline-asm"]


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def _newer(a, b):
    return not os.path.exists(b) or os.path.getmtime(a) > os.path.getmtime(b)


DEBUG_LIB = os.path.join(LIBDIR, "libprego_amd_debug.so")
DEBUG_ABI_SOURCES = ("miniroad.cpp", "vit_host.cpp")       # the only files that define prego_debug_* entry points (include/prego_amd_debug.h)
DEBUG_ONLY_SOURCES = ("debug_hog.hip",)                    # kernels of probe hooks: linked into libprego_amd_debug.so only


def _tree_hash() -> str:
    import hashlib
    h = hashlib.sha256()
    for f in sources() + sorted(x for x in os.listdir(CSRC) if x.endswith(".h")):
        h.update(open(os.path.join(CSRC, f), "rb").read())
    inc = os.path.join(os.path.dirname(HERE), "include")
    for f in sorted(os.listdir(inc)):
        h.update(open(os.path.join(inc, f), "rb").read())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = False) -> str:
    """Builds libprego_amd.so (the product ABI, include/prego_amd.h) and libprego_amd_debug.so (the same objects plus the probe /
    unit-test entry points of include/prego_amd_debug.h: the two host files compiled again with -DPREGO_DEBUG_ABI).  An object is
    recompiled when it is older than its source or a header; EVERYTHING is rebuilt when the library's recorded sources_sha256 differs
    from the tree (a prebuilt library shipped with an edited tree, or clock skew between hosts, cannot pass as current)."""
    os.makedirs(LIBDIR, exist_ok=True)
    info_path = os.path.join(LIBDIR, "build_info.json")
    stale = True
    if os.path.exists(info_path) and os.path.exists(LIB) and os.path.exists(DEBUG_LIB):
        try:
            import json
            stale = json.load(open(info_path)).get("sources_sha256") != _tree_hash()
        except Exception:
            stale = True
    force = force or (stale and os.path.exists(info_path))       # hash mismatch of an existing build: rebuild every object
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    inc = os.path.join(os.path.dirname(HERE), "include")
    hdrs += [os.path.join(inc, f) for f in os.listdir(inc)]
    objs, dbg_objs, jobs = [], [], []
    for src in sources():
        sp = os.path.join(CSRC, src)
        op = os.path.join(LIBDIR, src.replace(".", "_") + ".o")
        if src not in DEBUG_ONLY_SOURCES:
            objs.append(op)
        variants = [(op, [])]
        if src in DEBUG_ABI_SOURCES:
            dop = os.path.join(LIBDIR, src.replace(".", "_") + "_dbg.o")
            variants.append((dop, ["-DPREGO_DEBUG_ABI"]))
            dbg_objs.append(dop)
        else:
            dbg_objs.append(op)
        for o, extra in variants:
            if force or _newer(sp, o) or any(_newer(h, o) for h in hdrs):
                jobs.append([HIPCC] + FLAGS + extra + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", sp, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if jobs or stale or not os.path.exists(LIB) or not os.path.exists(DEBUG_LIB):
        run([HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs)
        run([HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", DEBUG_LIB] + dbg_objs)
        _write_build_info(len(jobs))
    return LIB


def _write_build_info(n_compiled: int):
    """provenance of the shared library next to it (prego_amd/lib/build_info.json, travels with the .so): where and when it was
    built, by which compiler, from which tree - `build_info()` / bench.py report it, so a run can tell a library built on its own box
    from one shipped prebuilt"""
    import json
    import platform
    import time
    try:
        ver = subprocess.run([HIPCC, "--version"], capture_output=True, text=True).stdout.splitlines()[0]
    except Exception:
        ver = "unknown"
    try:
        git = subprocess.run(["git", "-C", os.path.dirname(HERE), "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
    except Exception:
        git = None
    info = {"built_at": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()), "host": platform.node(), "hipcc": ver, "arch": ARCH,
            "flags": FLAGS, "sources_sha256": _tree_hash(), "git_head_at_build": git, "objects_compiled_in_this_call": n_compiled,
            "sources": len(sources())}
    json.dump(info, open(os.path.join(LIBDIR, "build_info.json"), "w"), indent=1)


def build_info() -> dict:
    """build_info.json of the library in use + whether it matches the sources in this tree and was built on this host"""
    import json
    import platform
    p = os.path.join(LIBDIR, "build_info.json")
    if not os.path.exists(p):
        return {"build_mode": "unknown (no build_info.json beside the library)"}
    info = json.load(open(p))
    info["sources_match_tree"] = _tree_hash() == info.get("sources_sha256")
    same_host = info.get("host") == platform.node()
    info["build_mode"] = ("built on this host" if same_host else "prebuilt elsewhere, shipped with the tree") + \
        ("" if info["sources_match_tree"] else " - STALE: sources differ from the tree it was built from")
    return info


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
