"""CPU: the C-ABI library loads and exports every symbol include/eegldm.h declares (no compute
calls without a GPU), the ctypes signature table covers the header, and the product package never
imports the oracle."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "synthetic-sleep-eeg-signal-generation-using-latent-diffusion-models_amd")


def header_symbols():
    src = open(os.path.join(ROOT, "include", "eegldm.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(eegldm_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(os.path.join(PKG, "libeegldm.so"))
    syms = header_symbols()
    assert len(syms) > 60
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, f"declared in eegldm.h but not exported: {missing}"
    assert lib.eegldm_abi_version() == 8


def test_ctypes_table_covers_header():
    from eegldm._lib import SIGNATURES
    syms = set(header_symbols())
    assert syms == set(SIGNATURES), (sorted(syms - set(SIGNATURES)), sorted(set(SIGNATURES) - syms))


def test_error_path_without_gpu():
    """Argument validation and the error string work without touching the device."""
    from eegldm._lib import lib
    lib.eegldm_last_error.restype = ctypes.c_char_p
    rc = lib.eegldm_ctx_sync(None)
    assert rc != 0 and b"null ctx" in lib.eegldm_last_error()


def test_product_never_imports_oracle():
    bad = []
    for dirpath, _d, files in os.walk(PKG):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M) or "oracle/" in txt:
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_no_cpu_fallback_without_gpu():
    """With no HIP device the product refuses to run (no silent CPU path)."""
    import pytest
    import torch
    import eegldm
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU fallback|MI355X"):
        eegldm.Context(0)


def test_no_packed_fp32_instructions_in_the_device_code():
    """the low lane of v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32 came out wrong in waves sharing a CU with the LDS-DMA GEMMs
    of another stream; the Makefile builds with -fno-slp-vectorize -fno-vectorize and this keeps it that way (a flag lost in a refactoring
    brings back ~41 000 of them in 450 kernels)."""
    import os
    import sys
    import pytest
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import check_packed_f32 as C
    if not (os.path.exists(C.LLVM + "/llvm-objdump") and os.path.exists(C.LLVM + "/llvm-objcopy")):
        pytest.skip("no llvm-objdump / llvm-objcopy here")
    lib = os.path.join(root, "synthetic-sleep-eeg-signal-generation-using-latent-diffusion-models_amd", "libeegldm.so")
    hist = C.packed_f32_by_kernel(lib)
    assert sum(hist.values()) == 0, dict(hist.most_common(10))
