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
