"""Operator-level view of one eager training step (torch.profiler): which ATen ops own the device time that is not in
lora_amd kernels.  Run on the GPU box:  python scripts/op_profile.py [cfg3] > gpurun_out/op_profile.txt
``cfg3``: BASELINE configs[3] (extended injection, rank 16, 768^2, batch 1, channels-last) and, for the copy / fill / sum
launches, the lora_amd source line that issued them (round 6: 110 copies + 60 fills + 22 sums per step in the adapter step that
its frozen twin does not have)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import lora_amd.lora as L  # noqa: E402
from lora_amd import trainer as T  # noqa: E402
from lora_amd.standin import DDPMScheduler  # noqa: E402

CFG3 = len(sys.argv) > 1 and sys.argv[1] == "cfg3"
dev = torch.device("cuda", 0)
torch.manual_seed(0)
unet = bench.build_unet(dev, torch.bfloat16, seed=0)
if CFG3:
    os.environ.setdefault("LORA_AMD_HEAD_PAD", "1")
    unet.to(memory_format=torch.channels_last)
    L.inject_trainable_lora_extended(unet, r=16)
else:
    L.inject_trainable_lora(unet, r=4)
T.promote_lora_to_fp32(unet)
unet.train()
state = T.FlatLoraState([{"params": T.lora_params(unet), "lr": 1e-4, "weight_decay": 1e-2}], max_grad_norm=1.0, device=dev)
state.attach_direct_grads(unet)
merged = state.enable_merged_weights(unet) if CFG3 else None
sched, cfg = DDPMScheduler(), T.StepConfig()
B, hw = (1, 96) if CFG3 else (4, 64)
lat = (torch.randn(B, 4, hw, hw, device=dev) * 0.18215).to(torch.bfloat16)
if CFG3:
    lat = lat.contiguous(memory_format=torch.channels_last)
ehs = torch.randn(B, 77, 768, device=dev).to(torch.bfloat16)


def step():
    T.forward_backward(unet, sched, lat, ehs, cfg, merged=merged)
    state.step(state.all_reduce())


for _ in range(3):
    step()
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=CFG3) as prof:
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

if CFG3:
    # who issues the copies / fills / sums: innermost lora_amd (or bench / standin) frame of the op's python stack
    from collections import Counter

    src = Counter()
    dev_us = Counter()
    for ev in prof.events():
        if ev.name in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::sum", "aten::contiguous", "aten::clone", "aten::to"):
            frame = next((f for f in (ev.stack or []) if "lora_amd" in f or "bench.py" in f), "?")
            key = (ev.name, frame.strip()[-110:], str(ev.input_shapes)[:70])
            src[key] += 1
            dev_us[key] += ev.self_device_time_total
    print("\n# copy / fill / sum launches by issuing source line (2 steps): count, device us, op, frame, shapes")
    for key, c in sorted(src.items(), key=lambda kv: -dev_us[kv[0]])[:60]:
        print(f"{c:5d} {dev_us[key]:9.1f}  {key[0]:16s} {key[1]}  {key[2]}")
