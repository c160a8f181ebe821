"""``loss_step`` of the PTI CLI (reference cli_lora_pti.py:260-370): the oracle restatement and ``lora_amd.cli_lora_pti``
against vectors produced by EXECUTING the reference's own function (scripts/make_golden.py::pti_loss_cases):
``t_mutliplier`` 0.8, both prediction types, the masked-MSE branch with a temperature, parameter gradients."""
import os
import types

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from lora_amd import cli_lora_pti as pti
from lora_amd.standin import DDPMScheduler
from oracle import torch_ref as TR
from tests import helpers as H

CASES = ["plain", "tmul", "mask", "mask_temp", "vpred_mask"]


@pytest.fixture(scope="module")
def gold():
    return dict(np.load(os.path.join(H.GOLDEN, "pti_loss_cases.npz")))


def _models(gold, device="cpu"):
    unet, text = H.PtiToyUNet(), H.PtiToyText()
    unet.load_state_dict({k: torch.from_numpy(gold["unet_" + k.replace(".", "_")]) for k in unet.state_dict()})
    text.emb.weight.data.copy_(torch.from_numpy(gold["text_emb"]))
    return unet.to(device), text.to(device)


def _case(gold, tag):
    tmul, temp, vpred, seed = gold[f"{tag}_meta"]
    mask = torch.from_numpy(gold[f"{tag}_mask"]) if f"{tag}_mask" in gold else None
    return (torch.from_numpy(gold[f"{tag}_latents"]), torch.from_numpy(gold[f"{tag}_ids"]),
            torch.from_numpy(gold[f"{tag}_noise"]), torch.from_numpy(gold[f"{tag}_t"]), mask, float(tmul), float(temp),
            "v_prediction" if vpred else "epsilon", int(seed))


@pytest.mark.parametrize("tag", CASES)
def test_oracle_pti_loss_matches_reference_vectors(gold, tag):
    lat, ids, noise, ts, mask, tmul, temp, ptype, _ = _case(gold, tag)
    unet, text = _models(gold)
    assert int(ts.max()) < int(1000 * tmul)  # the reference drew inside [0, 1000 * t_mutliplier)
    loss = TR.pti_loss_step(lambda x, t, c: unet(x, t, c).sample, lat, noise, ts, text(ids)[0],
                            DDPMScheduler().alphas_cumprod, mask, temp, ptype)
    loss.backward()
    assert abs(loss.item() - float(gold[f"{tag}_loss"])) <= 1e-6 * max(1.0, abs(float(gold[f"{tag}_loss"])))
    np.testing.assert_allclose(unet.conv.weight.grad.numpy(), gold[f"{tag}_dconv"], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(unet.ctx.weight.grad.numpy(), gold[f"{tag}_dctx"], rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("tag", CASES)
def test_loss_step_matches_reference_vectors_cpu(gold, tag):
    """The product's loss_step, same seed as the reference run: same draws (noise, then timesteps), same loss, same
    gradients — including the masked branch and the 0.8 timestep range."""
    lat, ids, _, _, mask, tmul, temp, ptype, seed = _case(gold, tag)
    unet, text = _models(gold)
    batch = {"pixel_values": lat, "input_ids": ids}
    if mask is not None:
        batch["mask"] = mask
    torch.manual_seed(seed)
    loss = pti.loss_step(batch, unet, None, text, DDPMScheduler(prediction_type=ptype), t_mutliplier=tmul,
                         mask_temperature=temp, cached_latents=True)
    loss.backward()
    assert abs(loss.item() - float(gold[f"{tag}_loss"])) <= 1e-6 * max(1.0, abs(float(gold[f"{tag}_loss"])))
    np.testing.assert_allclose(unet.conv.weight.grad.numpy(), gold[f"{tag}_dconv"], rtol=1e-4, atol=1e-7)


def test_loss_step_vae_branch_matches_reference_vector(gold):
    class ToyVAE:  # as in scripts/make_golden.py::pti_loss_cases
        def encode(self, px):
            lat = F.avg_pool2d(px, 8)[:, [0, 1, 2, 0]] * torch.tensor([1.0, -0.5, 0.25, 2.0]).view(1, 4, 1, 1)
            return types.SimpleNamespace(latent_dist=types.SimpleNamespace(sample=lambda: lat))

    unet, text = _models(gold)
    batch = {"pixel_values": torch.from_numpy(gold["vae_pixels"]), "input_ids": torch.from_numpy(gold["vae_ids"])}
    torch.manual_seed(2000)
    loss = pti.loss_step(batch, unet, ToyVAE(), text, DDPMScheduler(), t_mutliplier=1.0, cached_latents=False)
    assert abs(loss.item() - float(gold["vae_loss"])) <= 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["tmul", "mask_temp", "vpred_mask"])
def test_loss_step_on_device_vs_oracle(gold, tag):
    """loss_step on the GPU (device-side draws) against the oracle on the host fed with the SAME draws, recovered by
    re-seeding the device generator and repeating loss_step's two calls; f32, tolerance 1e-5 relative."""
    lat, ids, _, _, mask, tmul, temp, ptype, seed = _case(gold, tag)
    dev = torch.device("cuda")
    unet, text = _models(gold, dev)
    batch = {"pixel_values": lat.to(dev), "input_ids": ids.to(dev)}
    if mask is not None:
        batch["mask"] = mask.to(dev)
    torch.manual_seed(seed)
    loss = pti.loss_step(batch, unet, None, text, DDPMScheduler(prediction_type=ptype), t_mutliplier=tmul,
                         mask_temperature=temp, cached_latents=True)
    loss.backward()
    torch.manual_seed(seed)
    noise = torch.randn_like(batch["pixel_values"])
    ts = torch.randint(0, int(1000 * tmul), (lat.shape[0],), device=dev).long()
    assert int(ts.max()) < int(1000 * tmul)
    cu, ct = _models(gold)
    ref = TR.pti_loss_step(lambda x, t, c: cu(x, t, c).sample, lat, noise.cpu(), ts.cpu(), ct(ids)[0],
                           DDPMScheduler().alphas_cumprod, mask, temp, ptype)
    ref.backward()
    assert abs(loss.item() - ref.item()) <= 1e-5 * max(1.0, abs(ref.item()))
    np.testing.assert_allclose(unet.conv.weight.grad.cpu().numpy(), cu.conv.weight.grad.numpy(), rtol=2e-3, atol=1e-6)
