"""Operator-level view of one eager training step (torch.profiler): which ATen ops own the device time that is not in
lora_amd kernels.  Run on the GPU box:  python scripts/op_profile.py > gpurun_out/op_profile.txt"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import lora_amd.lora as L  # noqa: E402
from lora_amd import trainer as T  # noqa: E402
from lora_amd.standin import DDPMScheduler  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
unet = bench.build_unet(dev, torch.bfloat16, seed=0)
L.inject_trainable_lora(unet, r=4)
T.promote_lora_to_fp32(unet)
unet.train()
state = T.FlatLoraState([{"params": T.lora_params(unet), "lr": 1e-4, "weight_decay": 1e-2}], max_grad_norm=1.0, device=dev)
state.attach_direct_grads(unet)
sched, cfg = DDPMScheduler(), T.StepConfig()
lat = (torch.randn(4, 4, 64, 64, device=dev) * 0.18215).to(torch.bfloat16)
ehs = torch.randn(4, 77, 768, device=dev).to(torch.bfloat16)


def step():
    T.forward_backward(unet, sched, lat, ehs, cfg)
    state.step(state.all_reduce())


for _ in range(3):
    step()
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(2):
        step()
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=False).table(sort_by="self_cuda_time_total", row_limit=45,
                                                           max_name_column_width=60))
print("\n# pointwise / copy operators by input shape (2 steps): name, calls, self device us, shapes")
rows = [e for e in prof.key_averages(group_by_input_shape=True)
        if e.key in ("aten::copy_", "aten::add", "aten::add_", "aten::mul", "aten::cat", "aten::sum", "aten::fill_",
                     "aten::zero_", "aten::silu", "aten::silu_backward", "aten::native_layer_norm",
                     "aten::native_layer_norm_backward", "aten::constant_pad_nd", "aten::upsample_nearest2d",
                     "aten::upsample_nearest2d_backward", "aten::div", "aten::mse_loss", "aten::normal_")]
rows.sort(key=lambda e: -e.self_device_time_total)
for e in rows[:120]:
    print(f"{e.key:36s} {e.count:5d} {e.self_device_time_total:10.1f}  {e.input_shapes}")
