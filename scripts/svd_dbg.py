import os, sys, copy, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lora_amd as L
from lora_amd import _C, cli_svd as S
from tests import helpers as H
for name in ("colreduce_batched", "rowdot_batched", "chol_inverse_batched"):
    orig = getattr(_C, name)
    def wrap(*a, _o=orig, _n=name, **k):
        out = _o(*a, **k)
        bad = torch.isnan(out).any().item() or torch.isinf(out).any().item()
        if bad or os.environ.get("V"):
            print(_n, tuple(a[0].shape), "->", tuple(out.shape), "nan/inf" if bad else "ok", "in_nan", torch.isnan(a[0]).any().item(), flush=True)
        return out
    setattr(_C, name, wrap)
Holder = H.named_class("CrossAttention")
def tree(seed):
    torch.manual_seed(seed)
    t = Holder()
    for nm in ("to_q", "to_k", "to_v"):
        t.add_module(nm, torch.nn.Linear(64, 48, bias=False))
    t.add_module("to_out", torch.nn.Linear(64, 96, bias=False))
    return t
base = tree(0); tuned = copy.deepcopy(base)
g = torch.Generator().manual_seed(1)
for m in tuned.children():
    u, v = torch.randn(m.out_features, 5, generator=g), torch.randn(5, m.in_features, generator=g)
    m.weight.data += (u * torch.tensor([1.0, 0.5, 0.25, 0.12, 0.002])) @ v * 0.05
ours_b, ours_t = copy.deepcopy(base).cuda(), copy.deepcopy(tuned).cuda()
L.inject_trainable_lora(ours_b, r=4); L.inject_trainable_lora(ours_t, r=4)
S.overwrite_base(ours_b, ours_t, rank=4, clamp_quantile=1.0)
for name in ("to_q", "to_k", "to_v", "to_out"):
    m = getattr(ours_b, name)
    res = (getattr(tuned, name).weight.data - getattr(base, name).weight.data).float()
    U, Sg, Vh = torch.linalg.svd(res, full_matrices=False)
    ref = (U[:, :4] * Sg[:4]) @ Vh[:4]
    prod = (m.lora_up.weight.data @ m.lora_down.weight.data).cpu()
    print(name, "nan", torch.isnan(prod).any().item(), "rel err", float((prod - ref).norm() / ref.norm()), "S", Sg[:6].tolist())
