#!/usr/bin/env python3
"""Registers, scratch and LDS of the kernels inside a built library (the notes of its gfx950 code object).
Usage: tools/kernel_regs.py [name-filter-regex] [library]"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def kernels(lib):
    """{mangled kernel name: {"vgpr", "sgpr", "scratch", "lds"}} over EVERY code object of the library (one offload bundle per
    translation unit, behind one another in .hip_fatbin)."""
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    notes = ""
    with tempfile.TemporaryDirectory() as t:
        fat = os.path.join(t, "fat.bin")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, lib, os.path.join(t, "discard.so")])
        blob = open(fat, "rb").read()
        starts = [i for i in range(len(blob)) if blob.startswith(magic, i)]
        for n, a in enumerate(starts):
            part, co = os.path.join(t, "part%d.bin" % n), os.path.join(t, "dev%d.co" % n)
            with open(part, "wb") as f:
                f.write(blob[a:starts[n + 1] if n + 1 < len(starts) else len(blob)])
            subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + part,
                                   "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
            if os.path.getsize(co):
                notes += subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", co]).decode() + "\n"
    out, cur = {}, None
    keys = {".private_segment_fixed_size": "scratch", ".vgpr_count": "vgpr", ".sgpr_count": "sgpr", ".group_segment_fixed_size": "lds"}
    block = {}
    for line in notes.splitlines():
        t = line.strip()
        if t.startswith("- .") or t.startswith("- "):           # a new kernel's record begins
            if "name" in block: out[block["name"]] = block
            block = {}
            t = t[2:].strip()
        if ":" in t:
            k, v = t.split(":", 1)
            if k == ".name": block["name"] = v.strip()
            elif k in keys: block[keys[k]] = int(v)
    if "name" in block: out[block["name"]] = block
    return out


if __name__ == "__main__":
    pat = re.compile(sys.argv[1] if len(sys.argv) > 1 else ".")
    lib = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "g-vom_amd", "lib", "libgvom_hip.so")
    ks = kernels(lib)
    try:
        names = subprocess.run(["c++filt"], input="\n".join(ks).encode(), stdout=subprocess.PIPE).stdout.decode().splitlines()
    except OSError:
        names = list(ks)
    for raw, nice in sorted(zip(ks, names), key=lambda p: p[1]):
        if pat.search(nice) and "vgpr" in ks[raw]:
            b = ks[raw]
            print("%-70s vgpr %3d sgpr %3d scratch %3d lds %6d" % (nice.split("(")[0][:70], b.get("vgpr", -1), b.get("sgpr", -1), b.get("scratch", -1), b.get("lds", -1)))
