"""Lists the packed-fp32 VALU instructions (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32) left in a build of libeegldm.so, by kernel.
Why it matters: DESIGN.md 3.3 -- their low lane is not reliable in a wave that shares a CU with the LDS-DMA GEMMs of another stream, so the
library is built with -fno-slp-vectorize and tests/test_abi.py keeps the count at zero.      python tools/check_packed_f32.py [lib.so]"""
import collections, os, re, struct, subprocess, sys, tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(lib):
    """The gfx950 code objects inside the .hip_fatbin section (clang offload bundles, one per translation unit)."""
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.run([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", lib, os.path.join(td, "copy")], check=True, capture_output=True)
        d = open(fat, "rb").read()
    pos, out = 0, []
    while True:
        i = d.find(MAGIC, pos)
        if i < 0:
            return out
        (ne,) = struct.unpack_from("<Q", d, i + 24)
        p = i + 32
        for _ in range(ne):
            off, size, tl = struct.unpack_from("<QQQ", d, p); p += 24
            triple = d[p:p + tl].decode(); p += tl
            if "amdgcn" in triple and size:
                out.append(d[i + off:i + off + size])
        pos = i + 24


def packed_f32_by_kernel(lib):
    hist = collections.Counter()
    with tempfile.TemporaryDirectory() as td:
        for k, co in enumerate(code_objects(lib)):
            f = os.path.join(td, f"co{k}.elf"); open(f, "wb").write(co)
            txt = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--demangle", f], capture_output=True, text=True).stdout
            cur = "?"
            for line in txt.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
                if m: cur = m.group(1)
                elif re.search(r"\bv_pk_(fma|add|mul)_f32\b", line): hist[cur] += 1
    return hist


if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "synthetic-sleep-eeg-signal-generation-using-latent-diffusion-models_amd", "libeegldm.so")
    h = packed_f32_by_kernel(lib)
    print(f"{lib}: {sum(h.values())} packed-fp32 instructions in {len(h)} kernels")
    for k, v in h.most_common(40): print(f"  {v:5d}  {k[:150]}")
