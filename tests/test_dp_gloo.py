"""N>1 path on CPU: world_size-2 gloo ranks with sharded batches must reproduce the single-process
large-batch step (the reference's DDP semantics: grads averaged over ranks, then clip, then AdamW)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import helpers as H


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build():
    import lora_amd as L
    from lora_amd import trainer as T
    from lora_amd.standin import tiny_unet

    torch.manual_seed(0)
    u = tiny_unet()
    u.requires_grad_(False)
    torch.manual_seed(5)
    L.inject_trainable_lora(u, r=2)
    for up, _ in L.extract_lora_ups_down(u):
        up.weight.data.normal_(0, 0.05)
    st = T.FlatLoraState([{"params": T.lora_params(u), "lr": 1e-3, "weight_decay": 1e-2}], max_grad_norm=1.0)
    u.train()
    return u, st


def _data(it):
    g = torch.Generator().manual_seed(100 + it)
    return (torch.randn(4, 4, 16, 16, generator=g), torch.randn(4, 7, 32, generator=g),
            torch.randn(4, 4, 16, 16, generator=g), torch.randint(0, 1000, (4,), generator=g))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, H.REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from lora_amd import trainer as T
    from lora_amd.standin import DDPMScheduler

    torch.set_num_threads(1)
    r, _, w = T.init_distributed("cpu")
    assert (r, w) == (rank, world) and dist.get_backend() == "gloo"
    u, st = _build()
    sched = DDPMScheduler()
    for it in range(3):
        lat, ehs, noise, t = _data(it)
        sl = slice(rank * 2, rank * 2 + 2)  # each rank owns half of the global batch
        T.forward_backward(u, sched, lat[sl], ehs[sl], T.StepConfig(), noise=noise[sl], timesteps=t[sl])
        scale = st.all_reduce()
        assert scale == 0.5
        st.step(scale)
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), st.flat_p.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_gloo_ranks_equal_single_process_large_batch(tmp_path):
    from lora_amd import trainer as T
    from lora_amd.standin import DDPMScheduler

    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    p0, p1 = np.load(tmp_path / "rank0.npy"), np.load(tmp_path / "rank1.npy")
    assert np.array_equal(p0, p1), "ranks diverged"
    u, st = _build()
    sched = DDPMScheduler()
    for it in range(3):
        lat, ehs, noise, t = _data(it)
        T.forward_backward(u, sched, lat, ehs, T.StepConfig(), noise=noise, timesteps=t)  # mean over the global batch
        st.step(st.all_reduce())
    # Adam normalises by sqrt(v): f32 summation-order noise in tiny grads moves a few elements by ~1e-5
    np.testing.assert_allclose(p0, st.flat_p.numpy(), rtol=1e-3, atol=3e-5)
