"""Lists the gfx950 kernels inside libdifformer_hip.so (no GPU needed): every `__CLANG_OFFLOAD_BUNDLE__` in the library's
.hip_fatbin section is unpacked, the gfx950 code object's symbol table is read with llvm-readelf, and every `<name>.kd` (kernel
descriptor) is one kernel instantiation.
    python scripts/kernel_symbols.py [--mangled] [path/to/lib.so]      -> one demangled kernel name per line, sorted
Used by scripts/kernel_coverage.py and tests/test_kernel_coverage.py."""
import os
import shutil
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_LIB = os.path.join(ROOT, "difformer_amd", "lib", "libdifformer_hip.so")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _tool(name):
    for p in ("/opt/rocm/lib/llvm/bin/" + name, shutil.which(name) or ""):
        if p and os.path.exists(p):
            return p
    raise RuntimeError(f"{name} not found")


def code_objects(lib=DEFAULT_LIB):
    """-> list of bytes: the gfx950 ELF code objects embedded in the shared library."""
    blob = open(lib, "rb").read()
    out, pos = [], 0
    while True:
        pos = blob.find(MAGIC, pos)
        if pos < 0:
            break
        n = struct.unpack_from("<Q", blob, pos + len(MAGIC))[0]
        q = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, q)
            triple = blob[q + 24:q + 24 + tlen].decode()
            q += 24 + tlen
            if "gfx950" in triple and size:
                out.append(blob[pos + off:pos + off + size])
        pos += len(MAGIC)
    return out


def mangled_kernels(lib=DEFAULT_LIB):
    names = set()
    readelf = _tool("llvm-readelf")
    with tempfile.TemporaryDirectory() as td:
        for i, co in enumerate(code_objects(lib)):
            p = os.path.join(td, f"co{i}.elf")
            open(p, "wb").write(co)
            txt = subprocess.run([readelf, "-s", "-W", p], capture_output=True, text=True, check=True).stdout
            for line in txt.splitlines():
                f = line.split()
                if len(f) >= 8 and f[-1].endswith(".kd"):
                    names.add(f[-1][:-3])
    return sorted(names)


def demangle(names):
    filt = shutil.which("c++filt") or _tool("llvm-cxxfilt")
    r = subprocess.run([filt], input="\n".join(names), capture_output=True, text=True, check=True).stdout.splitlines()
    return [normalize(x) for x in r]


def normalize(name):
    """One spelling for a kernel name whether it comes from c++filt or from rocprofv3's kernel trace: no anonymous-namespace
    prefix, no parameter list, no `void ` return type, no spaces."""
    s = name.strip().replace("(anonymous namespace)::", "")
    if s.startswith("void "):
        s = s[5:]
    depth, cut = 0, len(s)
    for i, ch in enumerate(s):           # cut the parameter list: the first '(' outside template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            cut = i
            break
    return s[:cut].replace(" ", "")


def kernels(lib=DEFAULT_LIB):
    return sorted(set(demangle(mangled_kernels(lib))))


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    lib = args[0] if args else DEFAULT_LIB
    ks = mangled_kernels(lib) if "--mangled" in sys.argv else kernels(lib)
    print("\n".join(ks))
    print(f"# {len(ks)} kernels", file=sys.stderr)
