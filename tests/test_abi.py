"""The C-ABI library loads without a GPU and exports every symbol include/pcompanion_hip.h
declares; the ctypes table covers the header.  No compute calls here."""
import ctypes
import os
import re

from conftest import ROOT


def header_functions():
    txt = open(os.path.join(ROOT, "include", "pcompanion_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(pc_[a-z0-9_]+)\s*\(", txt)))


def test_header_declares_the_path():
    fns = header_functions()
    for must in ("pc_p2v_ffn_forward_train", "pc_p2v_ffn_backward", "pc_p2v_attention_forward",
                 "pc_p2v_attention_backward", "pc_p2v_triplet_loss", "pc_adam_step", "pc_p2v_train_step",
                 "pc_build_similarity_batch", "pc_mt_negative_samples", "pc_joint_forward", "pc_joint_loss",
                 "pc_joint_backward", "pc_joint_train_step", "pc_gather_rows", "pc_scatter_add_rows"):
        assert must in fns


def test_library_exports_every_declared_symbol():
    from p_companion_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "run python -m p_companion_amd.build (hipcc cross-compiles gfx950)"
    L = ctypes.CDLL(_lib.LIB_PATH)
    for name in header_functions():
        assert hasattr(L, name), f"{name} declared in the header but not exported"
    assert L.pc_abi_version() == 8
    assert L.pc_build_flags() == 0, "a developer-knob build (PC_EXP_* / *_TIMING) is not the product"


def test_ctypes_table_matches_header():
    from p_companion_amd import _lib
    assert sorted(_lib.SIGNATURES) == header_functions()
    _lib.lib()          # binds every signature; raises on a missing symbol


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from p_companion_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    import pytest
    with pytest.raises(_lib.HipKernelError):
        _lib.lib()


def test_ops_refuse_cpu_tensors():
    """No CPU fallback: CPU tensors are rejected before any kernel is touched."""
    import pytest
    import torch
    from p_companion_amd import ops
    with pytest.raises(TypeError):
        ops.linear_forward(torch.zeros(4, 128), torch.zeros(8, 128))
    with pytest.raises(TypeError):
        ops.gather_rows(torch.zeros(4, 128), torch.zeros(2, dtype=torch.int32))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "p_companion_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "oracle/" not in src, f


def test_exchange_slot_plumbing_without_a_gpu():
    """ABI 6 on the host side only (no kernel runs): the exchange slot's trampoline delivers (pointer, count, stream) and turns a
    Python exception into PC_ECOMM + a re-raise; pc_exchange_adam validates before anything is launched; the RCCL entry points
    resolve from the copy of librccl torch has loaded; no exchange is made for a single process."""
    import pytest
    from p_companion_amd import _lib, distributed as pdist, ops
    L = _lib.lib()
    seen = []
    ex = ops.CallbackExchange(lambda ptr, n, stream: seen.append((ptr, n, stream)))
    fn = ops.EXCHANGE_FN(ex.fn.value)
    assert fn(ex.ctx, ctypes.c_void_p(0x1000), 7, ctypes.c_void_p(0x20)) == 0 and seen == [(0x1000, 7, 0x20)]

    def boom(ptr, n, stream):
        raise KeyError("no such buffer")

    bad = ops.CallbackExchange(boom)
    assert ops.EXCHANGE_FN(bad.fn.value)(bad.ctx, None, 1, None) == -5
    with pytest.raises(KeyError):
        bad.reraise()
    bad.reraise()                                                     # (delivered once)
    # argument checks come before any launch: NULL buffers, and the device-counter form without its scratch scalars
    assert L.pc_exchange_adam(None, None, None, None, None, None, 8, None, 1, None, 1e-3, 0.9, 0.999, 1e-8, None) == -1
    assert L.pc_rccl_allreduce_mean(None, None, 0, None) == -1 and L.pc_rccl_comm_destroy(None) == -1
    assert L.pc_rccl_available() in (0, 1) and isinstance(L.pc_rccl_last_error(), bytes)
    if L.pc_rccl_available():
        buf = ctypes.create_string_buffer(128)
        assert L.pc_rccl_unique_id(None) == -1
    assert pdist.make_exchange(1) is None


def test_no_shipped_kernel_spills_vector_registers():
    """The code objects' own metadata (llvm-readelf --notes on the gfx950 images of the compiled units): no kernel keeps vector
    registers in scratch, and the only private segments are the one-time generators' / the stand-alone batch builder's."""
    from p_companion_amd import build
    build.build()
    res = build.kernel_resources()
    assert len(res) > 100
    assert not [(n, r["vgpr_spill_count"]) for n, r in res.items() if r["vgpr_spill_count"] > 0]
    allowed = ("gen_coview_kernel", "gen_complementary_kernel", "build_complementary_batch_kernel")
    assert all(any(a in n for a in allowed) for n, r in res.items() if r["private_segment_fixed_size"] > 0)
    # the two persistent GEMM families at the occupancies their launches are sized for (registers per lane: 512 / waves per SIMD)
    for n, r in res.items():
        if "gemm_nt_kernelILi2ELi2ELi16ELi3E" in n:
            assert r["vgpr_count"] <= 168, (n, r["vgpr_count"])          # three workgroups of four waves per CU
