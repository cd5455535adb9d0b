"""Registers, spills and LDS of every kernel in the built library, read from the code objects' own metadata.

    python tools/kernel_resources.py [pattern ...]

libva_hip.so carries one clang offload bundle per translation unit; the gfx950 entry of a bundle is an ELF whose
NT_AMDGPU_METADATA note (msgpack) lists, per kernel, what the compiler allocated.  tests/test_kernel_resources.py holds
the default kernels to "no spilled vector registers" with it (a spill in the one-wave row pipeline once halved the
1280x720 throughput without failing any test)."""
import os
import struct
import subprocess
import sys

import msgpack

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "video_analytics_amd", "libva_hip.so")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(path=LIB):
    data = open(path, "rb").read()
    i = data.find(MAGIC)
    while i >= 0:
        n = struct.unpack_from("<Q", data, i + 24)[0]
        off = i + 32
        for _ in range(n):
            o, s, tl = struct.unpack_from("<QQQ", data, off)
            off += 24
            triple = data[off:off + tl].decode()
            off += tl
            if "gfx950" in triple and s:
                yield data[i + o:i + o + s]
        i = data.find(MAGIC, i + 1)


def elf_notes(elf):
    assert elf[:4] == b"\x7fELF" and elf[4] == 2, "64-bit ELF expected"
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum = struct.unpack_from("<HH", elf, 0x3A)
    for k in range(shnum):
        sh = shoff + k * shentsize
        sh_type, = struct.unpack_from("<I", elf, sh + 4)
        if sh_type != 7:  # SHT_NOTE
            continue
        off, size = struct.unpack_from("<QQ", elf, sh + 0x18)
        p, end = off, off + size
        while p + 12 <= end:
            namesz, descsz, ntype = struct.unpack_from("<III", elf, p)
            p += 12
            name = elf[p:p + namesz].rstrip(b"\0")
            p += (namesz + 3) & ~3
            desc = elf[p:p + descsz]
            p += (descsz + 3) & ~3
            yield name, ntype, desc


def demangle(names):
    for tool in ("/opt/rocm/lib/llvm/bin/llvm-cxxfilt", "c++filt"):
        try:
            out = subprocess.run([tool] + names, capture_output=True, text=True, check=True).stdout.split("\n")
            return [o.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "") for o in out[:len(names)]]
        except Exception:
            continue
    return names


def kernels(path=LIB):
    """-> list of dicts: name (demangled, without the argument list), vgpr, agpr, sgpr, vgpr_spill, sgpr_spill, lds, scratch, max_wg"""
    rows = []
    for elf in code_objects(path):
        for name, ntype, desc in elf_notes(elf):
            if name == b"AMDGPU" and ntype == 32:
                md = msgpack.unpackb(desc, raw=False, strict_map_key=False)
                for k in md.get("amdhsa.kernels", []):
                    rows.append(dict(mangled=k[".name"], vgpr=k.get(".vgpr_count", 0), agpr=k.get(".agpr_count", 0), sgpr=k.get(".sgpr_count", 0),
                                     vgpr_spill=k.get(".vgpr_spill_count", 0), sgpr_spill=k.get(".sgpr_spill_count", 0),
                                     lds=k.get(".group_segment_fixed_size", 0), scratch=k.get(".private_segment_fixed_size", 0),
                                     max_wg=k.get(".max_flat_workgroup_size", 0)))
    for r, n in zip(rows, demangle([r["mangled"] for r in rows])):
        r["name"] = n
    return rows


if __name__ == "__main__":
    pats = sys.argv[1:]
    print("%-64s %5s %5s %5s %7s %7s %7s %6s" % ("kernel", "vgpr", "agpr", "sgpr", "v-spill", "s-spill", "lds", "wg"))
    for r in sorted(kernels(), key=lambda r: r["name"]):
        if not pats or any(p in r["name"] for p in pats):
            print("%-64s %5d %5d %5d %7d %7d %7d %6d" % (r["name"][:64], r["vgpr"], r["agpr"], r["sgpr"], r["vgpr_spill"], r["sgpr_spill"], r["lds"], r["max_wg"]))
