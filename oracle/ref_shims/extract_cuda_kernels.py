"""ORACLE build recipe: pull the `__global__` kernel definitions out of a reference .cu file, verbatim, into a generated
include under oracle/_ref/ (git-ignored; never committed).

The reference's CUDA sources cannot be compiled here (no nvcc) and their launchers use `<<<...>>>`, but the four
ransac_voting kernels are plain C inside: one thread per output element, no shared memory, no atomics, no barriers.
Compiled for the host behind a grid emulator (ransac_ref_shim.cpp) they give the reference's own arithmetic as the
golden source for row a10.  Difference to a real CUDA build worth knowing: nvcc contracts a*b+c into FMAs by default,
g++ -O2 on x86-64 does not — the oracle, the HIP kernels (-ffp-contract=off) and this build all use the uncontracted form."""
import re
import sys


def extract(src: str) -> str:
    out = []
    for m in re.finditer(r"__global__", src):
        start = m.start()
        brace = src.index("{", start)
        depth, i = 0, brace
        while True:
            c = src[i]
            if c == "{":
                depth += 1
            elif c == "}":
                depth -= 1
                if depth == 0:
                    break
            i += 1
        out.append(src[start:i + 1])
    return "\n\n".join(out) + "\n"


if __name__ == "__main__":
    with open(sys.argv[1]) as f:
        text = extract(f.read())
    with open(sys.argv[2], "w") as f:
        f.write("// GENERATED at build time from %s by oracle/ref_shims/extract_cuda_kernels.py — do not commit\n" % sys.argv[1])
        f.write(text)
