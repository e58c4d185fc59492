"""Reads the gfx950 code object inside a hipcc-built shared library: which kernels it holds and whether any of them was
compiled for threadgroup-split mode (``-mtgsplit``).

Why it matters here: ``k_cascade_bulk`` reads back, with PLAIN loads, the survivor list its own workgroup stored one pass
earlier (``csrc/kernels_cascade.hpp``, ``MIRGE_SURV_PLAIN_LOADS``).  That is correct only while all waves of a workgroup
run on ONE compute unit and so share its vector L1 -- the default.  In tgsplit mode a workgroup's waves may sit on
different CUs, their L1s are not coherent, and a workgroup barrier no longer makes the stores visible to plain loads.  The
compiler offers no macro for the mode, so the build (``__graft_entry__.build``) and a CPU test look at what was actually
built: bit 16 (TG_SPLIT) of ``compute_pgm_rsrc3`` in every kernel descriptor.
"""
import struct

_BUNDLE_MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def device_elf(path, arch="gfx950"):
    """The bytes of the ``arch`` code object bundled into the shared library ``path``."""
    blob = open(path, "rb").read()
    at = blob.find(_BUNDLE_MAGIC)
    while at >= 0:
        n, = struct.unpack_from("<Q", blob, at + 24)
        p = at + 32
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if size and triple.endswith(arch) and blob[at + off:at + off + 4] == b"\x7fELF":
                return blob[at + off:at + off + size]
        at = blob.find(_BUNDLE_MAGIC, at + 1)
    raise RuntimeError(f"{path}: no {arch} code object found")


def kernel_descriptors(elf):
    """{kernel name: the 64 bytes of its descriptor} from the ``<name>.kd`` symbols of an AMDGPU ELF."""
    if elf[:4] != b"\x7fELF" or elf[4] != 2:
        raise RuntimeError("not a 64-bit ELF")
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum = struct.unpack_from("<HH", elf, 0x3A)
    secs = []
    for i in range(shnum):
        _, typ, _, addr, off, size, link, _, _, entsize = struct.unpack_from("<IIQQQQIIQQ", elf, shoff + i * shentsize)
        secs.append((typ, addr, off, size, link, entsize))
    out = {}
    for typ, _, off, size, link, entsize in secs:
        if typ not in (2, 11) or not entsize:  # SHT_SYMTAB, SHT_DYNSYM
            continue
        str_off = secs[link][2]
        for k in range(size // entsize):
            name_i, _, _, _, value, sz = struct.unpack_from("<IBBHQQ", elf, off + k * entsize)
            end = elf.index(b"\0", str_off + name_i)
            name = elf[str_off + name_i:end].decode()
            if not name.endswith(".kd") or sz != 64:
                continue
            for styp, saddr, soff, ssize, _, _ in secs:
                if styp != 8 and saddr <= value and value + 64 <= saddr + ssize:  # not SHT_NOBITS
                    out[name[:-3]] = elf[soff + value - saddr:soff + value - saddr + 64]
                    break
    return out


def tg_split_kernels(path, arch="gfx950"):
    """Names of the kernels of ``path`` whose descriptor asks for threadgroup-split mode (expected: none)."""
    kds = kernel_descriptors(device_elf(path, arch))
    if not kds:
        raise RuntimeError(f"{path}: the code object holds no kernel descriptor")
    return sorted(n for n, kd in kds.items() if struct.unpack_from("<I", kd, 44)[0] & (1 << 16))


def n_kernels(path, arch="gfx950"):
    return len(kernel_descriptors(device_elf(path, arch)))
