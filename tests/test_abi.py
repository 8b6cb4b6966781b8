"""The C-ABI library loads without a GPU and exports every symbol include/lnx.h declares."""
import ctypes as C
import os
import re

import pytest

from linnaeus_amd import _lib as L

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(REPO, "include", "lnx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(lnx_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_lists_agree():
    assert declared_symbols() == sorted(L.EXPORTS)


def test_library_loads_and_exports_everything():
    lib = L.lib()
    for name in declared_symbols():
        assert hasattr(lib, name), name
    assert lib.lnx_version() >= 100


def test_struct_layouts_match_header_sizes():
    # sizes the C compiler produces for the argument structs (LP64): catches field drift
    assert C.sizeof(L.RowMap) == 12
    assert C.sizeof(L.GemmArgs) % 8 == 0 and C.sizeof(L.WgradArgs) % 8 == 0
    assert C.sizeof(L.PrepDesc) == 56


def test_ctypes_mirrors_have_the_sizes_the_c_compiler_gives(tmp_path):
    """Every argument struct of include/lnx.h against its ctypes mirror: sizeof from a gcc-compiled probe (a field added to
    the header but not to the binding would make the library read past the Python object)."""
    import shutil
    import subprocess

    from linnaeus_amd.model import _Cfg

    if shutil.which("gcc") is None:
        pytest.skip("no C compiler")
    pairs = {
        "lnx_rowmap": L.RowMap, "lnx_gemm_args": L.GemmArgs, "lnx_wgrad_args": L.WgradArgs, "lnx_ln_args": L.LnArgs, "lnx_ln_bwd_args": L.LnBwdArgs,
        "lnx_dwconv_args": L.DwconvArgs, "lnx_dwconv_wgrad_args": L.DwconvWgradArgs, "lnx_attn_args": L.AttnArgs, "lnx_attn_bwd_args": L.AttnBwdArgs,
        "lnx_prep_desc": L.PrepDesc, "lnx_softce_args": L.SoftCEArgs, "lnx_mix_args": L.MixArgs, "lnx_adamw_desc": L.AdamWDesc,
        "lnx_adamw_hyper": L.AdamWHyper, "lnx_convmlp_args": L.ConvMlpArgs, "lnx_convmlp_bwd_args": L.ConvMlpBwdArgs,
        "lnx_mformer_cfg": _Cfg, "lnx_meta_head_args": L.MetaHeadArgs, "lnx_meta_head_bwd_args": L.MetaHeadBwdArgs,
        "lnx_rope_table": L.RopeTable,
    }
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include "lnx.h"\nint main(void) {\n' +
                   "".join(f'    printf("{n} %zu\\n", sizeof({n}));\n' for n in pairs) + "    return 0;\n}\n")
    exe = tmp_path / "sizes"
    subprocess.run(["gcc", "-I", os.path.join(REPO, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    got = dict(line.split() for line in out.strip().splitlines())
    for n, cls in pairs.items():
        assert int(got[n]) == C.sizeof(cls), (n, got[n], C.sizeof(cls))


def test_argument_validation_without_gpu():
    lib = L.lib()
    a = L.GemmArgs()
    a.dtype = 7
    assert lib.lnx_gemm_nt(C.byref(a), None) != 0
    assert b"dtype" in lib.lnx_last_error()
    a.dtype, a.M, a.N, a.K = L.BF16, 4, 4, 3  # K not a multiple of 8
    assert lib.lnx_gemm_nt(C.byref(a), None) != 0
    assert b"multiple" in lib.lnx_last_error()


def test_fp8_gemm_rejects_the_activation_forms_it_does_not_carry():
    """lnx_gemm_nt_fp8 / _mxfp8 carry NONE, GELU (+ pre-activation copy) and GELU_BWD.  GELU_D / MUL_AUX / ReLU map onto the same
    epilogue feature masks as those, so they must be refused by name (they would silently compute something else, or load from a
    NULL aux).  Checked on the host, before any launch: the pointers below are never dereferenced."""
    lib = L.lib()
    fake = C.c_void_p(0x1000)
    for act in (L.ACT_GELU_D, L.ACT_MUL_AUX, L.ACT_RELU, L.ACT_RELU_BWD):
        a = L.GemmArgs()
        a.dtype, a.M, a.N, a.K = L.BF16, 256, 256, 256
        a.A, a.lda, a.W, a.ldw, a.C, a.ldc = fake, 256, fake, 256, fake, 256
        a.act = act
        assert lib.lnx_gemm_nt_fp8(C.byref(a), None, None, None) != 0, act
        assert b"not carried by the fp8 kernels" in lib.lnx_last_error(), lib.lnx_last_error()
        assert lib.lnx_gemm_nt_mxfp8(C.byref(a), fake, fake, None) != 0, act
        assert b"not carried by the fp8 kernels" in lib.lnx_last_error(), lib.lnx_last_error()


def test_conv_mlp_scratch_size_comes_from_the_launcher():
    """lnx_convmlp_bwd_ws_floats (what the plan sizes the fused-LayerNorm scratch with) = 2 C floats per workgroup of the launch the
    library would make: resident-weight kernels (C <= 96) at most 256 workgroups of 8 waves x 16 (32 at C = 32) rows, streamed-weight
    kernels one workgroup per 128 rows; unsupported widths report 0."""
    lib = L.lib()
    assert lib.lnx_convmlp_bwd_ws_floats(96, 256 * 56 * 56) == 256 * 2 * 96
    assert lib.lnx_convmlp_bwd_ws_floats(96, 1000) == -(-1000 // 128) * 2 * 96
    assert lib.lnx_convmlp_bwd_ws_floats(32, 1000) == -(-1000 // 256) * 2 * 32
    assert lib.lnx_convmlp_bwd_ws_floats(100, 1000) == 0 and lib.lnx_convmlp_bwd_ws_floats(96, 0) == 0
    if not os.environ.get("LNX_CM_NW"):
        assert lib.lnx_convmlp_bwd_ws_floats(192, 256 * 28 * 28) == (256 * 28 * 28 // 128) * 2 * 192


def test_cu_margin_is_validated():
    lib = L.lib()
    assert lib.lnx_set_cu_margin(-1) != 0 and b"lnx_set_cu_margin" in lib.lnx_last_error()
    assert lib.lnx_set_cu_margin(5000) != 0
    assert lib.lnx_set_cu_margin(0) == 0


def test_plan_create_validates_and_enumerates_parameters():
    from linnaeus_amd.model import _Cfg

    lib = L.lib()
    lib.lnx_plan_param_name.restype = C.c_char_p
    lib.lnx_plan_param_numel.restype = C.c_int64
    lib.lnx_plan_workspace_bytes.restype = C.c_int64
    cfg = _Cfg()
    cfg.dtype, cfg.batch, cfg.img_h, cfg.img_w, cfg.in_chans = L.BF16, 2, 224, 224, 3
    cfg.dims[:] = [96, 192, 384, 768]
    cfg.conv_depths[:] = [3, 3]
    cfg.rope_depths[:] = [5, 2]
    cfg.rope_heads[:] = [6, 12]
    cfg.mlp_hidden[:] = [1536, 3072]
    cfg.n_meta = 2
    cfg.meta_dims[0], cfg.meta_dims[1] = 2, 3
    cfg.n_tasks = 0
    h = C.c_void_p()
    assert lib.lnx_plan_create(C.byref(cfg), C.byref(h)) == 0, lib.lnx_last_error()
    n = lib.lnx_plan_num_params(h)
    assert n == 225  # 225 state_dict tensors for sm without heads
    total = sum(lib.lnx_plan_param_numel(h, i) for i in range(n))
    assert total == 29_189_091
    assert lib.lnx_plan_num_drop_calls(h) == 6 + 2 * 7
    assert lib.lnx_plan_workspace_bytes(h) > 0
    lib.lnx_plan_destroy(h)
    cfg.rope_heads[0] = 5  # head_dim != 64
    assert lib.lnx_plan_create(C.byref(cfg), C.byref(h)) != 0
    assert b"head_dim" in lib.lnx_last_error()
    cfg.rope_heads[0] = 6
    cfg.img_h = 230
    assert lib.lnx_plan_create(C.byref(cfg), C.byref(h)) != 0
