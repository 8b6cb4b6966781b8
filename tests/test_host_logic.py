"""Host-side mirror of the reference's model interface (no GPU needed): registry/build_model
behaviour, state_dict contract for all four architectures, config plumbing, loud failure on CPU."""
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
