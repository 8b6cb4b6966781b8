"""Host-side mirror of the reference's model interface (no GPU needed): registry/build_model
behaviour, state_dict contract for all four architectures, config plumbing, loud failure on CPU."""
import os

import pytest
import torch

import linnaeus_amd
from linnaeus_amd import arch_config, build_model, default_config
from linnaeus_amd.registry import _model_registry, create_model, register_model
from oracle import mformer_oracle as O
from tests.cases import CASES, TinyTree, make_config


def test_unknown_model_type_raises_value_error():
    cfg = default_config()
    cfg.MODEL.TYPE = "nope"
    with pytest.raises(ValueError, match="Unknown model type"):
        create_model(cfg)


def test_reregistration_overwrites():
    orig = _model_registry["mFormerV1"]

    @register_model("mFormerV1")
    class Other(torch.nn.Module):
        def __init__(self, config, **kw):
            super().__init__()

    assert _model_registry["mFormerV1"] is Other
    register_model("mFormerV1")(orig)
    assert _model_registry["mFormerV1"] is orig


@pytest.mark.parametrize("name", ["tiny_a", "tiny_b", "sm"])
def test_state_dict_names_shapes_order(name):
    spec = CASES[name]
    img = {"tiny_a": 64, "tiny_b": 96, "sm": 224}[name]
    model = build_model(make_config(spec, img), num_classes={t: c for t, c in spec.heads})
    shapes = O.param_shapes(spec)
    sd = model.state_dict()
    assert list(sd.keys()) == list(shapes.keys())
    assert all(tuple(sd[k].shape) == shapes[k] for k in shapes)


def test_hierarchical_heads_share_classifiers_and_register_hmatrices():
    spec = CASES["tiny_c"]
    tasks = [t for t, _ in spec.heads]
    nc = {t: c for t, c in spec.heads}
    tree = TinyTree({"taxa_L10": {0: 0, 1: 0, 2: 1, 3: 1, 4: 2, 5: 2}, "taxa_L20": {0: 0, 1: 0, 2: 1}}, tasks, nc)
    model = build_model(make_config(spec, 64, "ConditionalClassifier"), num_classes=nc, taxonomy_tree=tree)
    keys = list(model.state_dict().keys())
    # head.{task}.level_classifiers.{task'}.* aliased under every task + hmatrix buffers
    assert "head.taxa_L10.level_classifiers.taxa_L30.weight" in keys
    assert "head.taxa_L20.hmatrix_taxa_L20_taxa_L10" in keys
    a = model.head["taxa_L10"].level_classifiers["taxa_L20"].weight
    b = model.head["taxa_L30"].level_classifiers["taxa_L20"].weight
    assert a is b
    n_unique = sum(p.numel() for p in model.parameters())
    assert n_unique == sum(int(torch.tensor(s).prod()) for s in O.param_shapes(spec).values())
    with pytest.raises(TypeError):
        build_model(make_config(spec, 64, "ConditionalClassifier"), num_classes=nc, taxonomy_tree=object())


@pytest.mark.parametrize("arch,params", [("sm", 29_189_091)])
def test_arch_param_counts(arch, params):
    model = build_model(arch_config(arch, 224))
    assert sum(p.numel() for p in model.parameters()) == params
    assert model.extra_token_num == 3 and list(model.meta_components) == ["TEMPORAL", "SPATIAL"]
    md = model.parameter_groups_metadata
    assert md["stages"]["rope_freqs"] == ["freqs"] and "meta_" in md["heads"]["meta_heads"]


def test_invalid_configs_raise_like_the_reference():
    cfg = arch_config("sm", 224)
    cfg.MODEL.CONVNEXT_STAGES.DIMS = [96, 192, 384]
    with pytest.raises(ValueError, match="length 4"):
        build_model(cfg)
    cfg = arch_config("sm", 224)
    cfg.MODEL.ROPE_STAGES.DIMS = [256, 768]
    with pytest.raises(ValueError, match="must match RoPE dim"):
        build_model(cfg)
    cfg = arch_config("sm", 224)
    cfg.MODEL.DROP_RATE = 0.1
    m = build_model(cfg)  # builds (eval forwards are fine: dropout is the identity there); a training forward raises
    assert m.drop_rate == 0.1


def test_cpu_inputs_fail_loudly():
    model = build_model(arch_config("sm", 224))
    with pytest.raises(linnaeus_amd._lib.LnxError, match="no CPU path"):
        model(torch.zeros(1, 3, 224, 224), torch.zeros(1, 5))
    with pytest.raises(linnaeus_amd._lib.LnxError):
        model.head  # noqa: B018 (heads exist but are empty here)
        from linnaeus_amd.heads import LinearHead

        LinearHead(8, 4)(torch.zeros(2, 8))


def test_drop_path_schedule_matches_reference_linspace():
    model = build_model(arch_config("sm", 224))
    probs = [b.drop_prob for s in model.stages for b in s]
    ref = [x.item() for x in torch.linspace(0, 0.2, 13)]
    assert probs == pytest.approx(ref)


def test_default_grad_mode_is_autograd():
    """gradients flow through autograd unless a caller opts into the zero-copy mode -- torch DDP hooks
    and torch.autograd.grad() need AccumulateGrad to run (test_gpu_model.py::test_torch_ddp_wrap_default_grad_mode runs it)."""
    from linnaeus_amd.ddp import DataParallel

    model = build_model(make_config(CASES["tiny_a"], 64), num_classes={t: c for t, c in CASES["tiny_a"].heads})
    assert model.grad_mode == "autograd"
    DataParallel(model, broadcast=False)
    assert model.grad_mode == "direct"


def test_build_model_loads_local_pretrained(tmp_path):
    """models/build.py:94-103 + utils/checkpoint.py:513-700 for a local single-source checkpoint: "model" key, `module.` prefix,
    the target model's drop_params patterns (head., meta_, norm., downsample_layers. are NOT taken from the checkpoint)."""
    spec = CASES["tiny_a"]
    nc = {t: c for t, c in spec.heads}
    torch.manual_seed(1)
    src = build_model(make_config(spec, 64), num_classes=nc)
    path = tmp_path / "ckpt.pth"
    torch.save({"model": {"module." + k: v for k, v in src.state_dict().items()}, "epoch": 3}, path)
    cfg = make_config(spec, 64)
    cfg.MODEL.PRETRAINED = str(path)
    torch.manual_seed(2)
    dst = build_model(cfg, num_classes=nc)
    a, b = src.state_dict(), dst.state_dict()
    dropped = [k for k in a if k.startswith(("head.", "meta_", "downsample_layers.")) or ".norm." in k or k.startswith("norm.")]
    taken = [k for k in a if k not in dropped]
    assert taken and dropped
    for k in taken:
        if "norm" in k and not torch.equal(a[k], b[k]):
            continue  # LayerNorm weights initialise to the same constants either way
        assert torch.equal(a[k], b[k]), k
    assert any(not torch.equal(a[k], b[k]) for k in dropped if a[k].dtype.is_floating_point and a[k].numel() > 8)
    for bad in ({"PRETRAINED": "hf://org/repo/model.pth"}, {"PRETRAINED": str(path), "PRETRAINED_SOURCE": "metaformer"}):
        cfg2 = make_config(spec, 64)
        for k, v in bad.items():
            cfg2.MODEL[k] = v
        with pytest.raises((FileNotFoundError, NotImplementedError)):
            build_model(cfg2, num_classes=nc)


_REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(_REF), reason="build-container only: the reference is not on the GPU box")
def test_install_into_linnaeus_swaps_the_reference_registry():
    """the reference-side binding.  In a child process (so that nothing of `linnaeus` stays imported
    here): import linnaeus.models, build mFormerV1 through the REFERENCE's build_model with its yacs CfgNode and a real
    TaxonomyTree -> the reference class; call install_into_linnaeus(); build again through the same reference entry point ->
    the HIP-backed class, whose state_dict has exactly the reference model's keys, shapes and order."""
    import subprocess
    import sys
    import textwrap

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent("""
        import logging, sys, warnings
        warnings.filterwarnings("ignore"); logging.disable(logging.CRITICAL)
        import torch
        from yacs.config import CfgNode as CN
        from linnaeus.config import get_default_config
        from linnaeus.models import build_model
        from linnaeus.models import model_factory as mf
        from linnaeus.utils.config_utils import load_config, merge_configs
        from linnaeus.utils.taxonomy.taxonomy_tree import TaxonomyTree

        def cfg():
            c = get_default_config()
            arch = load_config("/root/reference/configs/model/archs/mFormerV1/mFormerV1_sm.yaml")
            c.MODEL = merge_configs(c.MODEL, arch.MODEL)
            c.MODEL.IMG_SIZE = 224
            tasks = ["taxa_L10", "taxa_L20", "taxa_L30"]
            c.DATA.TASK_KEYS_H5 = tasks
            c.MODEL.CLASSIFICATION.HEADS = CN({t: {"TYPE": "ConditionalClassifier", "ROUTING_STRATEGY": "soft", "TEMPERATURE": 1.0, "USE_BIAS": True} for t in tasks})
            return c, tasks

        c, tasks = cfg()
        nc = {"taxa_L10": 6, "taxa_L20": 3, "taxa_L30": 2}
        tree = TaxonomyTree({"taxa_L10": {0: 0, 1: 0, 2: 1, 3: 1, 4: 2, 5: 2}, "taxa_L20": {0: 0, 1: 0, 2: 1}}, tasks, nc)
        ref = build_model(c, num_classes=nc, taxonomy_tree=tree)
        assert type(ref).__module__.startswith("linnaeus."), type(ref)
        import linnaeus_amd
        from linnaeus_amd.registry import install_into_linnaeus
        assert install_into_linnaeus()
        ours = build_model(c, num_classes=nc, taxonomy_tree=tree)
        assert type(ours) is linnaeus_amd.model.mFormerV1 and mf._model_registry["mFormerV1"] is linnaeus_amd.model.mFormerV1, type(ours)
        a, b = ref.state_dict(), ours.state_dict()
        assert list(a.keys()) == list(b.keys()), [k for k in a if k not in b][:5] + [k for k in b if k not in a][:5]
        for k in a:
            assert a[k].shape == b[k].shape and a[k].dtype == b[k].dtype, k
        assert [n for n, _ in ref.named_parameters()] == [n for n, _ in ours.named_parameters()]
        ours.load_state_dict(a, strict=True)
        assert ref.parameter_groups_metadata == ours.parameter_groups_metadata
        assert ref.pretrained_ckpt_handling_metadata == ours.pretrained_ckpt_handling_metadata
        assert set(ours.head.keys()) == set(ref.head.keys()) and all(hasattr(h, "set_gradnorm_mode") for h in ours.head.values())
        print("OK", len(a))
    """)
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1",
               PYTHONPATH=os.pathsep.join([os.path.join(repo, "tests", "golden", "gen", "_stubs"), _REF, repo]))
    r = subprocess.run([sys.executable, "-c", code], cwd="/tmp", env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_plan_footprint_counts_dropout_masks_and_logits():
    """AutoBatch's predicted bytes include what lives outside the plan workspace -- the per-forward dropout
    keep masks (one byte per element of three [M, C] / [M, hidden] tensors per RoPE block, B x heads x N x Np for the attention
    probabilities) and the logits / dlogits buffers."""
    from linnaeus_amd.autobatch import predicted_bytes

    spec = CASES["tiny_a"]
    nc = {t: c for t, c in spec.heads}
    plain = build_model(make_config(spec, 64), num_classes=nc)
    cfg = make_config(spec, 64)
    cfg.MODEL.DROP_RATE, cfg.MODEL.ATTN_DROP_RATE = 0.2, 0.1
    drop = build_model(cfg, num_classes=nc)
    B = 4
    f0, f1 = plain.plan_footprint(B, 64), drop.plan_footprint(B, 64)
    assert f0["dropout"] == f0["attn_dropout"] == 0 and f0["workspace"] == f1["workspace"]
    want = attn = 0
    for s in range(2):
        N = (4 >> s) ** 2 + 3
        C_, hid, h = spec.rope_dims[s], int(spec.rope_dims[s] * spec.mlp_ratio[s]), spec.rope_heads[s]
        want += spec.rope_depths[s] * B * N * (2 * C_ + hid)
        attn += spec.rope_depths[s] * B * h * N * ((N + 63) // 64 * 64)
    assert f1["dropout"] == want and f1["attn_dropout"] == attn
    assert f1["logits"] == 2 * 4 * B * sum((c + 7) // 8 * 8 for c in nc.values())
    assert drop.plan_footprint(B, 64, train=False)["dropout"] == 0
    assert predicted_bytes(drop, B, 64) - predicted_bytes(plain, B, 64) == want + attn


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the GPU-less failure mode of the self-launcher")
def test_bench_gpus_n_launches_its_own_ranks():
    """`python bench.py --gpus 2` with WORLD_SIZE unset starts the ranks itself (torch.distributed.run as a
    child process, never exec) before any GPU call in the parent; in a GPU-less container the only failure is
    torch.cuda.set_device inside the children (the launcher stops the second rank as soon as the first has failed, so one or two
    tracebacks reach stderr), and the parent propagates the launcher's return code."""
    import subprocess
    import sys

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "torch.distributed.run" in r.stderr and r.stderr.count("torch.cuda.set_device(local)") in (1, 2), r.stderr[-3000:]
    assert "No HIP GPUs are available" in r.stderr


def test_bench_legs():
    """What `bench.py --gpus N` times: N = 1 is BASELINE config 2 (256 images; plus, untimed for `value`, config 3's 128 images on the one GPU); N > 1 times config 3's 128 images per GPU (global
    1024 at 8: the line's `value`) AND the 256-per-GPU weak leg in the same run; --batch overrides both."""
    import importlib.util

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(repo, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.legs_for(1, None) == [("config2", 256), ("config3_n1", 128)]  # round 5: the N = 1 line carries config 3's per-GPU shape too
    for n in (2, 4, 8):
        legs = bench.legs_for(n, None)
        assert legs[0] == ("config3", 128) and legs[0][1] * 8 == 1024 and ("weak256", 256) in legs
    assert bench.legs_for(8, 64) == [("batch", 64)]


@pytest.mark.skipif(not os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")), reason="needs hipcc (cross-compiles without a GPU)")
def test_counted_waits_match_the_compiled_store_counts():
    """gemm_nt_v9 / gemm_nt_v7 keep epilogue stores in flight by COUNT (`s_waitcnt vmcnt(pieces + stores)`); if the
    compiler ever emitted fewer store instructions than the source counts, a K slice would be read from LDS before it has landed.
    tools/audit_counted_waits.py compiles gemm5.hip / gemm3.hip for gfx950 and compares, per kernel instantiation, the 16-byte stores in
    the assembly with EpiStores / 2 x NSTORE (also run by __graft_entry__.build(): a mismatch fails the build)."""
    import importlib.util

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("audit_counted_waits", os.path.join(repo, "tools", "audit_counted_waits.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    ok, lines = mod.audit(arch="gfx950")
    assert ok, "\n".join(lines)
    assert sum("gemm_nt_v9" in ln for ln in lines) >= 6 and sum("gemm_nt_v7" in ln for ln in lines) >= 10, lines


def test_design_dispatch_table_is_what_the_library_decides():
    """DESIGN.md's NT dispatch table (section 6) is generated from lnx_nt_dispatch -- the decision function
    lnx_gemm_nt itself uses (gemm2.hip: nt_v2_family), callable without a GPU -- and this test regenerates it: a dispatch rule
    that changes without the document fails here.  Spot checks pin the rules the parity tests rely on."""
    import importlib.util

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("gen_dispatch_table", os.path.join(repo, "tools", "gen_dispatch_table.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    with open(os.path.join(repo, "DESIGN.md")) as fh:
        in_doc = gen.parse(fh.read())
    assert in_doc is not None, "DESIGN.md has no dispatch-table markers"
    assert in_doc.strip() == gen.table().strip(), "DESIGN.md section 6 is stale: python tools/gen_dispatch_table.py --write"
    from linnaeus_amd import _lib as L

    assert gen.query(256 * 199, 1536, 384, "fc1d") == L.NT_KERNEL_V9 and gen.query(256 * 199, 384, 384, "res_f32") == L.NT_KERNEL_V7
    assert gen.query(128 * 199, 1536, 384, "mul_aux") == L.NT_KERNEL_V7 and gen.query(128 * 199, 4096, 1024, "fc1d") == L.NT_KERNEL_V9
    assert gen.query(200, 1000, 768, "bias") == L.NT_KERNEL_SKINNY and gen.query(800, 384, 384, "plain") == L.NT_KERNEL_V1
