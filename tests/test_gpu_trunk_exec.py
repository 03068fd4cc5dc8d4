"""The fp16 trunk executors (trunk_exec.py on the MFMA convolution) against their float32 master modules on stock
convolutions, same weights, same inputs.  Bar: 1e-2 of the output range (dozens of fp16-rounded layers deep; the
discrete consumers of these outputs are tested end to end in test_gpu_vsr.py)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from video_super_resolution_amd.trunk_exec import FlowNet2Exec, HourglassExec, OSVOSExec  # noqa: E402


def _rel(a, ref):
    return (a - ref).abs().max().item() / ref.abs().max().item()


@pytest.mark.parametrize("hw", [(64, 96), (72, 88)])
def test_hourglass_exec(gpu_vsr, hw):
    netg = gpu_vsr.DepthModule.model.netG
    fr = torch.from_numpy(np.random.RandomState(1).randint(0, 256, (2,) + hw + (3,)).astype(np.float32)).cuda()
    with torch.no_grad():
        ref = netg(fr.permute(0, 3, 1, 2))
        got = HourglassExec(netg)(fr)
    assert got.shape == ref.shape
    assert _rel(got, ref) < 1e-2


@pytest.mark.parametrize("hw", [(70, 90), (64, 96), (135, 240)])
def test_hourglass_deferred_upsampling_is_bit_identical(gpu_vsr, hw):
    """`up` + AddResized as one pass (the x2 map never written, whichever arm ends in `up`) against the two passes: same index
    arithmetic, so the same bits -- including the levels whose sizes are odd (a 135-row skip resized onto a 2 x 67-row arm)."""
    netg = gpu_vsr.DepthModule.model.netG
    fr = torch.from_numpy(np.random.RandomState(4).randint(0, 256, (2,) + hw + (3,)).astype(np.float32)).cuda()
    ex = HourglassExec(netg)
    assert ex.defer_up
    with torch.no_grad():
        fused = ex(fr).clone()
        ex.defer_up = False
        try:
            two_pass = ex(fr).clone()
        finally:
            ex.defer_up = True
    assert torch.equal(fused, two_pass)


def test_flownet2_exec(gpu_vsr):
    net = gpu_vsr.FlowModule.net
    x = torch.from_numpy(np.random.RandomState(2).randint(0, 256, (2, 3, 2, 64, 128)).astype(np.float32)).cuda()
    with torch.no_grad():
        ref = net(x)
        got = FlowNet2Exec(net)(x)
    assert got.shape == ref.shape == (2, 2, 64, 128)
    # five cascaded sub-networks with warps in between, every layer rounding to fp16: the worst pixel may reach a few
    # percent of the flow range while the bulk stays far below
    assert _rel(got, ref) < 5e-2
    assert (got - ref).abs().mean().item() < 5e-3 * ref.abs().max().item()


def test_osvos_exec(gpu_vsr):
    net = gpu_vsr.VOSModule.net
    x = torch.from_numpy(np.random.RandomState(3).randint(0, 256, (2, 3, 70, 94)).astype(np.float32)).cuda() - 110.0
    with torch.no_grad():
        ref = net(x)
        got = OSVOSExec(net)(x)
    assert got.shape == ref.shape
    assert _rel(got, ref) < 1e-2


# ------------------------------------------------------------------------------------------------------------------
# The same executors against the REFERENCE's golden vectors (tests/golden/g4_wrappers.npz, written by the imported
# reference through oracle/make_golden.py), not only against this repository's own fp32 masters.  Bars = the measured
# errors recorded in DESIGN.md section 7 plus a margin; every test prints what it measured.
def test_flownet2_exec_vs_reference_golden(golden, gpu_vsr):
    g = golden("g4_wrappers")
    big = torch.from_numpy(g["flow_frames"]).cuda()              # [2,64,128,3]
    x = big.permute(3, 0, 1, 2).unsqueeze(0).contiguous()        # [1,3,2,64,128] (FlowProjectionModule.py:27-28)
    with torch.no_grad():
        got = FlowNet2Exec(gpu_vsr.FlowModule.net)(x).cpu()
    ref = torch.from_numpy(g["flow"])
    mx, mean = _rel(got, ref), (got - ref).abs().mean().item() / ref.abs().max().item()
    print(f"[FlowNet2Exec fp16 vs golden flow] max {mx:.3e} mean {mean:.3e} of range")
    assert mx < 4e-2 and mean < 4e-3      # measured 2.06e-2 / 2.27e-3


def test_hourglass_exec_vs_reference_golden(golden, gpu_vsr):
    g = golden("g4_wrappers")
    fr = torch.from_numpy(g["frames"]).cuda()                    # [2,32,48,3]
    with torch.no_grad():
        z = HourglassExec(gpu_vsr.DepthModule.model.netG)(fr)    # [2,1,32,48]
        got = gpu_vsr.DepthModule.combine(z[0:1], z[1:2]).cpu()  # DepthProjectionModule.py:14-18
    ref = torch.from_numpy(g["depth"])
    mx = _rel(got, ref)
    print(f"[HourglassExec fp16 vs golden depth] max {mx:.3e} of range")
    assert mx < 3e-3                       # measured 9.1e-4


def test_osvos_exec_vs_reference_golden(golden, gpu_vsr):
    g = golden("g4_wrappers")
    fr = torch.from_numpy(g["frames"]).cuda()
    x = (fr - gpu_vsr.VOSModule.meanval).permute(0, 3, 1, 2).contiguous()
    with torch.no_grad():
        got = OSVOSExec(gpu_vsr.VOSModule.net)(x).cpu()
    ref = torch.from_numpy(g["vos_logits"])
    mx = _rel(got, ref)
    s = torch.sigmoid(got[0, 0]) + torch.sigmoid(got[1, 0])
    flips = ((s > 0.7).float().numpy() != g["vos_mask"]).mean()
    print(f"[OSVOSExec fp16 vs golden logits] max {mx:.3e} of range, mask flips {flips:.4f}")
    assert mx < 4e-3 and flips < 5e-3      # measured 1.26e-3, no flipped mask pixel
