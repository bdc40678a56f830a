#!/usr/bin/env python3
"""Identity of the library a measurement was taken on: sha256 of g-vom_amd/lib/libgvom_hip.so, sha256 of the sources it is
built from (SOURCES below, in that order, so that anyone can recompute it from a
commit), and the git commit.  hipcc's output is reproducible: the same sources give the same library bytes.
The GPU box has no .git: `make -C g-vom_amd stamp` (run here, before gpurun) leaves the commit in g-vom_amd/lib/GIT_HEAD.

    tools/lib_identity.py                  print the identity as JSON
    tools/lib_identity.py --sidecar F      write F.meta.json beside a profile file F (kernel_stats.csv and the like)
"""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "g-vom_amd")
SOURCES = ["csrc/gvom_trace.hip", "csrc/gvom_fuse.hip", "csrc/gvom_map2d.hip", "csrc/gvom_stats.hip", "csrc/gvom_capi.hip", "csrc/gvom_comm.hip",
           "csrc/gvom_device.h", "csrc/gvom_internal.h", "../include/gvom_hip.h", "../include/gvom_hip_test.h", "Makefile"]


def sha256_file(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 20), b""):
            h.update(chunk)
    return h.hexdigest()


def source_sha256():
    h = hashlib.sha256()
    for rel in SOURCES:
        with open(os.path.join(PKG, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def git_head():
    try:
        out = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True, timeout=10)
        if out.returncode == 0 and out.stdout.strip():
            dirty = subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--", "g-vom_amd/csrc", "include", "g-vom_amd/Makefile"],
                                   capture_output=True, text=True, timeout=10).stdout.strip()
            return out.stdout.strip() + ("+uncommitted-source-changes" if dirty else "")
    except (OSError, subprocess.SubprocessError):
        pass
    try:
        with open(os.path.join(PKG, "lib", "GIT_HEAD")) as f:
            return f.read().strip() or None
    except OSError:
        return None


def identity(lib=None):
    lib = lib or os.path.join(PKG, "lib", "libgvom_hip.so")
    return {"lib_sha256": sha256_file(lib) if os.path.exists(lib) else None, "source_sha256": source_sha256(), "git_head": git_head(),
            "lib": os.path.relpath(lib, ROOT)}


def main():
    ident = identity()
    if len(sys.argv) > 2 and sys.argv[1] == "--sidecar":
        for f in sys.argv[2:]:
            with open(f + ".meta.json", "w") as o:
                json.dump(dict(ident, file=os.path.basename(f)), o, indent=1, sort_keys=True)
    else:
        print(json.dumps(ident, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
