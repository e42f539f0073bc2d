#!/usr/bin/env python3
"""What the shipped library's gfx950 code object says about every kernel: registers, spills, scratch, LDS.

The fat binary section (.hip_fatbin) of libsplat_hip.so is a clang offload bundle; its gfx950 entry is an ELF whose
NT_AMDGPU_METADATA note (msgpack) holds one record per kernel.  No ROCm tool is needed (roc-obj-ls wants a Perl module the
image lacks, llvm-objdump --offloading writes files next to the library): this parses both containers directly.

usage: python tools/codeobj.py [library] [name-substring ...]
"""
import struct
import sys

import msgpack

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
NT_AMDGPU_METADATA = 32


def _elf_sections(blob):
    assert blob[:4] == b"\x7fELF" and blob[4] == 2 and blob[5] == 1, "not a little-endian ELF64"
    shoff, = struct.unpack_from("<Q", blob, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", blob, 0x3A)
    secs = []
    for k in range(shnum):
        name, typ, _flags, _addr, off, size = struct.unpack_from("<IIQQQQ", blob, shoff + k * shentsize)
        secs.append((name, typ, off, size))
    stroff = secs[shstrndx][2]
    out = {}
    for name, typ, off, size in secs:
        end = blob.index(b"\0", stroff + name)
        out[blob[stroff + name:end].decode()] = (typ, off, size)
    return out


def code_objects(path):
    """{triple: bytes} of the offload bundle inside a host library."""
    blob = open(path, "rb").read()
    secs = _elf_sections(blob)
    if ".hip_fatbin" not in secs:
        raise RuntimeError("%s has no .hip_fatbin section" % path)
    _, off, size = secs[".hip_fatbin"]
    fat = blob[off:off + size]
    out = {}
    at = 0
    while True:                      # (one bundle per translation unit that was linked in)
        at = fat.find(MAGIC, at)
        if at < 0:
            break
        n, = struct.unpack_from("<Q", fat, at + len(MAGIC))
        p = at + len(MAGIC) + 8
        for _ in range(n):
            eoff, esize, tsize = struct.unpack_from("<QQQ", fat, p)
            triple = fat[p + 24:p + 24 + tsize].decode()
            p += 24 + tsize
            if esize:
                out.setdefault(triple, []).append(fat[at + eoff:at + eoff + esize])
        at += len(MAGIC)
    return out


def kernels(path, arch="gfx950"):
    """{kernel symbol (demangled name when the metadata has it): metadata dict} for the device code of `arch`."""
    found = {}
    for triple, objs in code_objects(path).items():
        if arch not in triple:
            continue
        for obj in objs:
            secs = _elf_sections(obj)
            for name, (typ, off, size) in secs.items():
                if typ != 7:         # SHT_NOTE
                    continue
                p = off
                while p < off + size:
                    namesz, descsz, ntype = struct.unpack_from("<III", obj, p)
                    p += 12
                    nname = obj[p:p + namesz]
                    p += (namesz + 3) & ~3
                    desc = obj[p:p + descsz]
                    p += (descsz + 3) & ~3
                    if ntype == NT_AMDGPU_METADATA and nname.startswith(b"AMDGPU"):
                        md = msgpack.unpackb(desc, raw=False, strict_map_key=False)
                        for k in md.get("amdhsa.kernels", []):
                            found[k.get(".name", "?")] = k
    return found


def demangle(sym):
    import subprocess
    try:
        return subprocess.run(["c++filt", sym], capture_output=True, text=True, check=True).stdout.strip()
    except Exception:
        return sym


def main():
    import os
    lib = sys.argv[1] if len(sys.argv) > 1 and os.path.exists(sys.argv[1]) else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "splat_amd", "libsplat_hip.so")
    pats = [a for a in sys.argv[1:] if not os.path.exists(a)]
    ks = kernels(lib)
    print("%-72s %5s %5s %6s %6s %7s %6s" % ("kernel", "vgpr", "sgpr", "vspill", "sspill", "scratch", "lds"))
    for sym in sorted(ks):
        k = ks[sym]
        name = demangle(sym).split("(")[0]
        if pats and not any(p in name for p in pats):
            continue
        print("%-72s %5d %5d %6d %6d %7d %6d" % (name[-72:], k[".vgpr_count"], k[".sgpr_count"], k.get(".vgpr_spill_count", 0),
                                                 k.get(".sgpr_spill_count", 0), k[".private_segment_fixed_size"], k[".group_segment_fixed_size"]))


if __name__ == "__main__":
    main()
