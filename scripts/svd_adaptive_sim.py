"""CPU (f64) model of cli_svd's subspace iteration — sketch 16, rank 8 — on spectra of different decay: per power iteration
the error of the rank-r product relative to the exact truncation's, and the two quantities the stopping rule of
``lora_amd/cli_svd.py`` looks at (gain of the top-r Ritz energy over the squared error, over the energy itself).
Behind DESIGN.md §3.6's statement that one or two iterations already give the Frobenius error within 0.6 % of optimal on every
spectrum tried, and that what does not settle on a flat spectrum is the individual vectors.  Run: python scripts/svd_adaptive_sim.py"""
import torch

torch.manual_seed(0)


def mk(N, K, s):
    n = min(N, K)
    U = torch.linalg.qr(torch.randn(N, n, dtype=torch.float64))[0]
    V = torch.linalg.qr(torch.randn(K, n, dtype=torch.float64))[0]
    return (U * s) @ V.T


def power_law(n, p=0.5):
    return torch.arange(1, n + 1, dtype=torch.float64) ** -p


def flat(n):
    return torch.sort(torch.ones(n, dtype=torch.float64) * (1 + 0.05 * torch.rand(n, dtype=torch.float64)), descending=True)[0]


def finetune_like(n):
    s = torch.full((n,), 0.3, dtype=torch.float64) * (1 + 0.2 * torch.rand(n, dtype=torch.float64))
    s[:8] = torch.tensor([6, 5, 4, 3, 2.2, 1.7, 1.3, 1.0], dtype=torch.float64)
    return torch.sort(s, descending=True)[0]


def planted(N, K, noise=1e-4):
    u, v = torch.randn(N, 12, dtype=torch.float64), torch.randn(12, K, dtype=torch.float64)
    sv = torch.tensor([3.0 * 0.6 ** i for i in range(12)], dtype=torch.float64)
    return (u / u.norm(dim=0)) @ torch.diag(sv) @ (v / v.norm(dim=1, keepdim=True)) * 0.2 + noise * torch.randn(N, K, dtype=torch.float64)


def run(A, r=8, l=16, iters=6):
    N, K = A.shape
    Q = torch.linalg.qr(A @ torch.randn(K, l, dtype=torch.float64))[0]
    U_, S_, Vh_ = torch.linalg.svd(A, full_matrices=False)
    best2 = float((S_[r:] ** 2).sum())
    norm2, prev, out = float((S_ ** 2).sum()), None, []
    for it in range(1, iters + 1):
        Qz = torch.linalg.qr(A.T @ Q)[0]
        Y = A @ Qz
        e = float(torch.linalg.eigvalsh(Y.T @ Y).flip(0)[:r].sum())
        Q = torch.linalg.qr(Y)[0]
        ub, s, vh = torch.linalg.svd(Q.T @ A, full_matrices=False)
        err2 = float((A - (Q @ ub[:, :r] * s[:r]) @ vh[:r]).norm() ** 2)
        gain = None if prev is None else e - prev
        out.append("it %d: err/best %.5f%s" % (it, (err2 / best2) ** 0.5, "" if gain is None else
                                                "  gain/err2 %.1e  gain/E %.1e" % (gain / max(norm2 - e, 1e-300), gain / e)))
        prev = e
    return out


if __name__ == "__main__":
    for name, A in (("sigma ~ i^-0.5, 320^2", mk(320, 320, power_law(320))), ("sigma ~ i^-0.25, 640^2", mk(640, 640, power_law(640, 0.25))),
                    ("flat, 640^2", mk(640, 640, flat(640))), ("fine-tune-like, 1280 x 320", mk(1280, 320, finetune_like(320))),
                    ("bench planted, 1280 x 2880", planted(1280, 2880)), ("bench planted, 10240 x 1280", planted(10240, 1280))):
        print(name)
        for ln in run(A):
            print("   ", ln)
