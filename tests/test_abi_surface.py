"""CPU-only: libog_decoder.so loads, exports every symbol include/og_decoder.h declares, and the
product path refuses to run without a GPU (no fallback)."""
import ctypes
import os
import re

import pytest
import torch

from offsetguided_amd import _lib, build as og_build, decoder

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "og_decoder.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(og_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(_lib.SIGNATURES)


def test_library_exports_every_declared_symbol():
    og_build.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(lib, name), f"{name} missing from libog_decoder.so"
    typed = _lib.load()
    assert typed.og_abi_version() == _lib.ABI_VERSION
    assert typed.og_topk_workspace_bytes(136, 640, 640, 32) > 0       # pure host arithmetic
    assert typed.og_group_workspace_bytes(8, 19, 32, 17, 128) == 8 * 128 * 17 * 6 * 4 + 8 * 19 * 32 * 11 * 4


def test_argument_validation_without_gpu():
    lib = _lib.load()
    rc = lib.og_nms_topk_f32(None, 1, 8, 8, 4, None, None, None, 0, None)
    assert rc == _lib.OG_EINVAL and b"null pointer" in lib.og_last_error()


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-only check")
def test_no_cpu_fallback():
    with pytest.raises(_lib.OgError):
        decoder.hmp_NMS(torch.zeros(1, 1, 8, 8))
    with pytest.raises(_lib.OgError):
        decoder.GreedyGroup(0.1).group_skeletons(__import__("numpy").zeros((19, 4, 13), "float32"))
