"""Is a maskless tiny-UNet step bit-reproducible call after call?  (tests/test_gpu_parity_r4.py::test_factor_pass_selection_modes_agree
saw two alternating values.)  Prints flat_g[:3] and a checksum for 10 identical calls, per factor-pass mode."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import lora_amd as L
from lora_amd import _C, ops, trainer as T
from lora_amd.standin import DDPMScheduler, tiny_unet
DEV = "cuda:0"
torch.manual_seed(0)
unet = tiny_unet(cross_attention_dim=64).to(DEV).to(torch.bfloat16)
unet.requires_grad_(False)
L.inject_trainable_lora(unet, r=4)
T.promote_lora_to_fp32(unet)
for m in unet.modules():
    if isinstance(m, L.LoraInjectedLinear):
        m.lora_up.weight.data.normal_(0, 0.05)
unet.train()
st = T.FlatLoraState([{"params": T.lora_params(unet), "lr": 1e-3, "weight_decay": 1e-2}], max_grad_norm=1.0, device=torch.device(DEV))
st.attach_direct_grads(unet)
merged = st.enable_merged_weights(unet)
g = torch.Generator().manual_seed(3)
lat, ehs = torch.randn(2, 4, 32, 32, generator=g).to(DEV).bfloat16(), torch.randn(2, 77, 64, generator=g).to(DEV).bfloat16()
noise, ts = torch.randn(2, 4, 32, 32, generator=g).to(DEV).bfloat16(), torch.randint(0, 1000, (2,), generator=g).to(DEV)
sched = DDPMScheduler()
for mode in ("masked", "all", "masked", "none"):
    _C.FACTORS_MFMA_MODE, _C.FACTORS_MFMA = mode, mode != "none"
    for i in range(6):
        st.zero_grad()
        loss = T.forward_backward(unet, sched, lat, ehs, T.StepConfig(), noise=noise, timesteps=ts, merged=merged)
        st.reduce_pending()
        print(mode, i, "loss %.7f" % float(loss), st.flat_g[:3].tolist(), "sum %.9e" % float(st.flat_g.double().sum()), flush=True)
