"""The reference's train step against the drop-in (main.py:205-213): the one differentiable SR call of VSR.forward
(video_super_resolution.py:64) is evaluated on stock differentiable operators when the module is in training mode and autograd
is on (sr.py:_forward_autograd, SURVEY.md 8(b) "autograd"); everything else -- eval mode, any call under no_grad -- runs the HIP
kernels.  Checked: same values as the kernels, gradients equal to the oracle's (CPU autograd over its stock-op restatement),
and the main.py step itself (fake MSE, `loss.data` overwritten, backward, Adam) updates the SR net only."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import vsr_oracle as O  # noqa: E402


def test_autograd_path_matches_kernels_and_oracle_gradients(gpu_vsr, oracle_params):
    m = copy.deepcopy(gpu_vsr.model)
    x = torch.from_numpy(np.random.RandomState(4).randint(0, 256, (8, 3, 10, 12)).astype(np.float32)).cuda()
    m.eval()
    ref_kernels = m(x)                                   # HIP kernels (fp32 configuration)
    m.train()
    with torch.no_grad():
        assert torch.equal(m(x), ref_kernels)            # training mode under no_grad still runs the kernels
    out = m(x)
    assert out.requires_grad and out.shape == ref_kernels.shape
    assert (out - ref_kernels).abs().max().item() <= 2e-5 * ref_kernels.abs().max().item()
    (out ** 2).mean().backward()
    P = {k[len("model."):]: v.detach().clone().requires_grad_(v.is_floating_point() and "mean" not in k)
         for k, v in oracle_params.items() if k.startswith("model.")}
    (O.sr_forward(P, x.cpu()) ** 2).mean().backward()
    checked = 0
    for name, p in m.named_parameters():
        if not p.requires_grad:
            assert p.grad is None                        # frozen MeanShift (blocks.py:54-55)
            continue
        g_ref = P[name].grad
        if g_ref is None:                                # upBlocks.5: hr[5] has no consumer (SRProjectionModule.py:54-80)
            assert p.grad is None or not p.grad.any(), name
            continue
        assert p.grad is not None and torch.isfinite(p.grad).all(), name
        scale = g_ref.abs().max().item()
        if scale > 0:
            assert (p.grad.cpu() - g_ref).abs().max().item() <= 2e-3 * scale, name
            checked += 1
    assert checked >= 60                                  # every conv / PReLU / Linear of the SR net with a live gradient


def test_reference_train_step_updates_only_the_sr_net(golden, gpu_vsr):
    g = golden("g10_loss")
    model = copy.deepcopy(gpu_vsr)
    model.train()                                         # main.py:178 (the guidance and loss networks stay in eval, vsr.py)
    assert model.model.training and not model.DepthModule.training and not model.SR_loss.training
    optimizer = torch.optim.Adam(model.parameters(), lr=1e-3)   # main.py:139
    from video_super_resolution_amd import driver
    data, target, high_frames = driver.ingest_item(torch.from_numpy(g["hr"]).unsqueeze(0).cuda(), 4)
    x, y, high_frame = data[0], target[0], high_frames[0]
    estimated_image = None
    total_loss = []
    optimizer.zero_grad()
    with torch.no_grad():                                 # main.py:199-203
        output, real_loss = model(x, y, high_frame, estimated_image)
        estimated_image = output
        total_loss.append(real_loss.data)
    before = {k: v.detach().clone() for k, v in model.state_dict().items()}
    output, real_loss = model(x, y, high_frame, estimated_image)                       # :206 (autograd on)
    loss = torch.nn.MSELoss()(output.cpu(), y.detach().clone().float().cpu())          # :207 fakeloss
    loss.data = sum(total_loss) / len(total_loss)                                       # :208
    loss.backward()                                                                     # :209
    optimizer.step()                                                                    # :210
    after = model.state_dict()
    changed = {k.split(".")[0] for k in before if not torch.equal(before[k], after[k])}
    assert changed == {"model"}, changed
    assert not torch.equal(before["model.conv_in.0.weight"], after["model.conv_in.0.weight"])
    assert torch.equal(before["model.sub_mean.bias"], after["model.sub_mean.bias"])
