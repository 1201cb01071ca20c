#!/usr/bin/env python3
"""Refuse to ship device code that holds the packed-fp32 form MI355X gets wrong beside another stream's MFMAs.

    python tools/check_isa_hazards.py gdrnpp_bop2022_amd/libgdrnpp_hip.so

Measured on MI355X (tools/pk_hazard_probe.py, profiles/r05p_pk_hazard_probe.txt): v_pk_add_f32 / v_pk_mul_f32 with op_sel:[0,1] (the
low result takes the HIGH half of src1) return wrong values in lanes 48..63 while another wave of the same SIMD issues
gfx950's double-rate 16-bit MFMAs (this library's GEMMs, f16 and bf16 32x32x16 alike; a bare v_mfma_f32_16x16x32_f16 loop in every
run) — kernels of two streams sharing the chip.  Plain packed forms and op_sel_hi-only forms passed every run.  hipcc emits the swizzled forms from the SLP vectorizer
only, so the library is built with -fno-slp-vectorize; this check disassembles what was actually linked (every gfx950 code
object of the .hip_fatbin section) and fails on ANY packed fp32 instruction with an explicit op_sel (conservative: op_sel:[1,0]
and [1,1] passed the probe)."""
import re
import shutil
import struct
import subprocess
import sys
import tempfile

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
BAD = re.compile(r"\bv_pk_(add|mul|fma)_f32\b.*\bop_sel:\[")


def tool(name):
    for cand in (f"/opt/rocm/lib/llvm/bin/{name}", shutil.which(name) or ""):
        if cand and shutil.os.path.exists(cand):
            return cand
    return None


def code_objects(lib_path):
    """gfx950 ELF images of a HIP shared library: the .hip_fatbin section is a sequence of clang offload bundles (one per
    translation unit): magic, u64 n, n x (u64 offset, u64 size, u64 len, triple)."""
    objcopy = shutil.which("objcopy") or tool("llvm-objcopy")
    if objcopy is None:
        raise RuntimeError("no objcopy")
    with tempfile.NamedTemporaryFile(suffix=".fatbin") as f:
        subprocess.run([objcopy, "-O", "binary", "--only-section=.hip_fatbin", lib_path, f.name], check=True)
        data = open(f.name, "rb").read()
    out, pos = [], 0
    while True:
        i = data.find(MAGIC, pos)
        if i < 0:
            return out
        n = struct.unpack_from("<Q", data, i + len(MAGIC))[0]
        p = i + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, p)
            p += 24
            triple = data[p:p + tl].decode()
            p += tl
            if "amdgcn" in triple and size:
                out.append((triple, data[i + off:i + off + size]))
        pos = i + len(MAGIC)


def scan(lib_path):
    """-> (number of code objects, number of packed fp32 instructions, [offending lines])."""
    objdump = tool("llvm-objdump")
    if objdump is None:
        raise RuntimeError("no llvm-objdump")
    n_pk, bad, objs = 0, [], code_objects(lib_path)
    for triple, blob in objs:
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(blob)
            f.flush()
            txt = subprocess.run([objdump, "-d", "--mcpu=" + triple.rsplit("-", 1)[-1], f.name], check=True, capture_output=True, text=True).stdout
        sym = "?"
        for line in txt.splitlines():
            if line.endswith(">:"):
                sym = line.split("<")[-1][:-2]
            elif "v_pk_" in line and "_f32" in line:
                n_pk += 1
                if BAD.search(line):
                    bad.append(f"{sym}: {line.split('//')[0].strip()}")
    return len(objs), n_pk, bad


def kernel_names(lib_path):
    """Demangled names of every kernel the library can launch (the ``<name>.kd`` kernel descriptors of its gfx950 code objects):
    the whitelist of tests/test_gpu_stream_guard.py — "a steady-state step launches nothing but these, copies and fills"."""
    readelf, cxxfilt = tool("llvm-readelf") or shutil.which("readelf"), tool("llvm-cxxfilt") or shutil.which("c++filt")
    if readelf is None or cxxfilt is None:
        raise RuntimeError("no (llvm-)readelf / c++filt")
    mangled = set()
    for _, blob in code_objects(lib_path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(blob)
            f.flush()
            txt = subprocess.run([readelf, "--symbols", "--wide", f.name], check=True, capture_output=True, text=True).stdout
        for line in txt.splitlines():
            name = line.split()[-1] if line.split() else ""
            if name.endswith(".kd"):
                mangled.add(name[:-3])
    out = subprocess.run([cxxfilt], input="\n".join(sorted(mangled)), check=True, capture_output=True, text=True).stdout
    return sorted(set(out.split("\n")) - {""})


def kernel_base_name(name):
    """'void gdrnpp::(anonymous namespace)::foo_kernel<1, 2>(float*, int)' / a mangled name -> 'foo_kernel'."""
    if name.startswith("_Z"):
        cxxfilt = tool("llvm-cxxfilt") or shutil.which("c++filt")
        if cxxfilt:
            name = subprocess.run([cxxfilt, name], check=True, capture_output=True, text=True).stdout.strip()
    name = name.replace("(anonymous namespace)::", "")
    cut = len(name)
    for i, ch in enumerate(name):           # drop the argument list and the template argument list
        if ch in "<(":
            cut = i
            break
    head = name[:cut].strip()
    head = head.split(" ")[-1]               # drop a leading return type ("void ")
    return head.split("::")[-1]


def main(argv):
    if len(argv) == 3 and argv[1] == "--kernels":
        for n in kernel_names(argv[2]):
            print(n)
        return 0
    if len(argv) != 2:
        print(__doc__)
        return 2
    n_obj, n_pk, bad = scan(argv[1])
    if n_obj == 0:
        print(f"check_isa_hazards: no device code found in {argv[1]}", file=sys.stderr)
        return 1
    if bad:
        print(f"check_isa_hazards: {len(bad)} packed-fp32 instruction(s) with op_sel in {argv[1]} (wrong in lanes 48..63 beside another "
              "stream's 16-bit MFMAs on MI355X; build with -fno-slp-vectorize):", file=sys.stderr)
        for b in bad[:20]:
            print("   " + b, file=sys.stderr)
        return 1
    print(f"check_isa_hazards: {n_obj} code objects, {n_pk} packed-fp32 instructions, none with op_sel")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
