"""Optimizer + step glue on device: FusedAdamW (gradient sum of squares in a fixed order, multi-tensor
AdamW with the clip folded in) against clip_grad_norm_ + torch.optim.AdamW, and on the reference's train-step fixture."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("clip", [None, 0.5, 1e6])
def test_fused_adamw_matches_torch(clip):
    from linnaeus_amd.optim import FusedAdamW

    g = torch.Generator().manual_seed(0)
    shapes = [(1000, 37), (5,), (4096,), (3, 3, 7, 7), (1,), (12289,)]
    pa = [torch.randn(*s, generator=g).cuda().requires_grad_(True) for s in shapes]
    pb = [p.detach().clone().requires_grad_(True) for p in pa]
    groups = lambda ps: [{"params": ps[:3], "lr": 3e-3, "weight_decay": 0.05}, {"params": ps[3:], "lr": 1e-2, "weight_decay": 0.0, "betas": (0.8, 0.95)}]
    oa = FusedAdamW(groups(pa), lr=1e-3, max_grad_norm=clip)
    ob = torch.optim.AdamW(groups(pb), lr=1e-3)
    for step in range(4):
        for a, b in zip(pa, pb):
            gr = torch.randn(a.shape, generator=g).cuda() * (0.1 + step)
            a.grad = gr.clone()
            b.grad = gr.clone()
        if step == 2:  # a scheduler changing the learning rate between steps
            oa.param_groups[0]["lr"] = ob.param_groups[0]["lr"] = 1e-3
        ref_norm = torch.nn.utils.clip_grad_norm_(pb, clip) if clip is not None else None
        oa.step()
        ob.step()
        if clip is not None:
            torch.testing.assert_close(oa.grad_norm(), ref_norm, rtol=1e-5, atol=1e-6)
        for a, b in zip(pa, pb):
            torch.testing.assert_close(a, b, rtol=2e-6, atol=2e-7)
    sd = oa.state_dict()
    assert set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"} and sd["state"][0]["step"] == 4
    for a, b in zip(pa, pb):
        torch.testing.assert_close(oa.state[a]["exp_avg_sq"], ob.state[b]["exp_avg_sq"], rtol=1e-5, atol=1e-9)


def test_clip_norm_is_the_same_bits_every_time():
    """Data-parallel replicas hold identical all-reduced gradients and must take identical steps: the gradient norm the clip
    reads is folded in a fixed order (it was a float atomicAdd per workgroup: last-bit differences between ranks, replicas
    drifting apart -- found by tests/test_gpu_ddp_two_ranks.py).  Many workgroups (6 M gradients), many repeats."""
    from linnaeus_amd.optim import FusedAdamW

    g = torch.Generator().manual_seed(3)
    ps = [torch.randn(n, generator=g).cuda().requires_grad_(True) for n in (4_000_003, 1_500_000, 777, 4096 * 100 + 1)]
    for p in ps:
        p.grad = torch.randn(p.shape, generator=g).cuda() * 3e-2
    want = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in ps))
    opt = FusedAdamW(ps, lr=0.0, weight_decay=0.0, max_grad_norm=1.0)  # (lr 0: the gradients and parameters stay as they are)
    seen = set()
    for _ in range(25):
        opt.step()
        seen.add(float(opt.grad_norm()))
    assert len(seen) == 1, seen
    assert abs(seen.pop() - float(want)) <= 2e-6 * float(want)


def test_fused_step_on_reference_train_fixture(golden_dir):
    """the reference's CE -> clip_grad_norm_(1.0) -> AdamW steps (tests/golden/train_step.npz) with the HIP model, the HIP
    loss and the HIP optimizer: every piece of the step on device"""
    from linnaeus_amd.loss import multitask_cross_entropy
    from linnaeus_amd.optim import FusedAdamW
    from tests.cases import load_train_step
    from tests.test_gpu_model import build

    spec, z, sd, x, meta, targets, weights = load_train_step(golden_dir)
    model = build("tiny_a", spec, sd, "fp32")
    model.train()
    model.grad_mode = "direct"
    xg, mg = x.cuda(), meta.cuda()
    tg = {t: v.cuda() for t, v in targets.items()}
    before = {k: v.detach().clone() for k, v in model.named_parameters()}
    opt = FusedAdamW(model.parameters(), lr=float(z["lr"]), weight_decay=float(z["wd"]), max_grad_norm=float(z["clip"]))
    for s in range(int(z["steps"])):
        model.zero_grad(set_to_none=True)
        loss = multitask_cross_entropy(model(xg, mg), tg, weights)
        loss.backward()
        opt.step()
        assert abs(loss.item() - float(z[f"loss_{s}"])) <= 2e-4 * abs(float(z[f"loss_{s}"]))
        assert abs(opt.grad_norm().item() - float(z[f"gnorm_{s}"])) <= 2e-3 * float(z[f"gnorm_{s}"])
    names = [str(n) for n in z["param_names"]]
    got = dict(model.named_parameters())
    for i, k in enumerate(names):
        if k == "aggregate.bias":
            continue  # exactly-zero gradient in real arithmetic (see test_gpu_model.test_train_step_matches_reference)
        d = (got[k].detach() - before[k]).double().norm().item()
        floor = 0.05 * float(z["lr"]) * int(z["steps"]) * got[k].numel() ** 0.5
        assert abs(d - z["delta_norms"][i]) <= 5e-2 * z["delta_norms"][i] + floor, (k, d, z["delta_norms"][i])


def test_fused_adamw_per_parameter_step_counts():
    """torch.optim.AdamW bias-corrects per parameter.  One parameter of a group skips a step (grad None),
    so its step count lags the others': the fused optimizer must still match torch exactly."""
    from linnaeus_amd.optim import FusedAdamW

    g = torch.Generator().manual_seed(3)
    pa = [torch.randn(257, generator=g).cuda().requires_grad_(True) for _ in range(3)]
    pb = [p.detach().clone().requires_grad_(True) for p in pa]
    oa, ob = FusedAdamW(pa, lr=1e-2, weight_decay=0.01), torch.optim.AdamW(pb, lr=1e-2, weight_decay=0.01)
    for step in range(5):
        for i, (a, b) in enumerate(zip(pa, pb)):
            if i == 1 and step in (1, 2):
                a.grad = b.grad = None
                continue
            gr = torch.randn(a.shape, generator=g).cuda()
            a.grad, b.grad = gr.clone(), gr.clone()
        oa.step()
        ob.step()
        for a, b in zip(pa, pb):
            torch.testing.assert_close(a, b, rtol=2e-6, atol=2e-7)
    assert oa.state[pa[1]]["step"] == 3 and oa.state[pa[0]]["step"] == 5


def test_muon_newton_schulz_and_steps_match_reference(golden_dir):
    """Muon (optimizers/muon.py) on the HIP GEMMs against the reference's own outputs (tests/golden/muon.npz): the
    Newton-Schulz orthogonalisation of wide / tall / non-multiple-of-8 / square matrices, and two optimizer steps on a
    wide, a tall and a conv-shaped parameter.  The reference rounds every product and affine term to bf16; the HIP path
    rounds once per fused launch, so it is compared both with the reference (bf16-level tolerance) and with the exact
    fp64 iteration (it must not be further from it than the reference is)."""
    from linnaeus_amd.optim import Muon, zeropower_via_newtonschulz5
    from oracle import mformer_oracle as O

    z = np.load(f"{golden_dir}/muon.npz")
    for name in ("wide", "tall", "odd", "sq"):
        G = torch.from_numpy(z[f"ns_in_{name}"])
        ref = torch.from_numpy(z[f"ns_out_{name}"])
        got = zeropower_via_newtonschulz5(G.cuda(), steps=5).float().cpu()
        exact = O.newton_schulz5_exact(G).float()
        assert got.shape == ref.shape
        e_ref, e_got = (ref - exact).abs().max().item(), (got - exact).abs().max().item()
        print(f"[muon/{name}] max|ref - exact| {e_ref:.4f}  max|hip - exact| {e_got:.4f}  max|hip - ref| {(got - ref).abs().max().item():.4f}")
        assert e_got <= 1.5 * e_ref + 2e-3, (name, e_got, e_ref)
        assert (got - ref).abs().max().item() <= 2.5 * e_ref + 4e-3, name
    params = [torch.nn.Parameter(torch.from_numpy(z[f"p{i}_init"]).cuda()) for i in range(3)]
    opt = Muon(params, lr=0.02, weight_decay=0.01, momentum=0.95, nesterov=True, ns_steps=5)
    for step in range(2):
        for i, p in enumerate(params):
            p.grad = torch.from_numpy(z[f"p{i}_grad{step}"]).cuda()
        opt.step()
    for i, p in enumerate(params):
        ref, init = torch.from_numpy(z[f"p{i}_final"]), torch.from_numpy(z[f"p{i}_init"])
        d_ref, d_got = ref - init, p.detach().cpu() - init
        rel = ((d_got - d_ref).norm() / d_ref.norm()).item()
        print(f"[muon/step p{i}] relative update error vs reference {rel:.4f}")
        assert rel <= 0.05, (i, rel)
    assert "momentum_buffer" in opt.state[params[0]]
