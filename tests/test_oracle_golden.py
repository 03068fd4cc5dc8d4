"""The oracle (oracle/vsr_oracle.py, oracle/native_ops.c) against the golden vectors captured from the
reference's own Python (oracle/make_golden.py).  CPU only.  The oracle restates fp32 arithmetic with the
same stock operators, so the bar here is (near) bit equality, far tighter than the 1e-3 of the device path."""
import numpy as np
import pytest
import torch

from oracle import vsr_oracle as O


def _sr_params(oracle_params):
    return {k[len("model."):]: v for k, v in oracle_params.items() if k.startswith("model.")}


@pytest.mark.parametrize("tag", ["16x16", "12x20"])
def test_sr_matches_reference(golden, oracle_params, tag):
    g = golden(f"g1_sr_{tag}")
    taps = {}
    with torch.no_grad():
        out = O.sr_forward(_sr_params(oracle_params), torch.from_numpy(g["x"]), taps=taps)
    assert out.shape == g["out"].shape
    np.testing.assert_array_equal(out.numpy(), g["out"])
    np.testing.assert_array_equal(taps["feat_in"].numpy(), g["feat_in"])
    for s in range(3):
        np.testing.assert_array_equal(taps[f"block{s}"].numpy(), g[f"block{s}"])
    np.testing.assert_array_equal(taps["prefc2"].numpy(), g["prefc2"])


def test_sr_group_dataflow_zero_fill(golden, oracle_params):
    """G2 pins D1: every lr[i]/hr[i] of the last step, as the reference computes them with empty->zeros."""
    g = golden("g2_groups")
    taps = {}
    with torch.no_grad():
        out = O.sr_forward(_sr_params(oracle_params), torch.from_numpy(g["x"]), taps=taps, group_taps_step=-1)
    np.testing.assert_array_equal(out.numpy(), g["out"])
    np.testing.assert_array_equal(taps["g_lr0"].numpy(), g["lr0"])
    for i in range(6):
        np.testing.assert_array_equal(taps[f"g_hr{i}"][0].numpy(), g[f"hr{i}"])
        np.testing.assert_array_equal(taps[f"g_lr{i + 1}"].numpy(), g[f"lr{i + 1}"])


def test_sr_input_independent_branches(golden):
    """Under zero fill lr[j] depends on the input only for j = 0 (mod 3): lr1, lr2, lr4, lr5 are identical for
    every image of the batch although the 8 images differ (the algebra the device path relies on)."""
    g = golden("g2_groups")
    for j in (1, 2, 4, 5):
        t = g[f"lr{j}"]
        assert np.array_equal(t[0], t[3]) and np.array_equal(t[0], t[7]), j
    for j in (3, 6):
        assert not np.array_equal(g[f"lr{j}"][0], g[f"lr{j}"][3])


def test_flow2img_matches_reference(golden):
    g = golden("g3_flow2img")
    for case in ("rand", "zero", "tiny", "nan_unknown", "unknown"):
        out = O.flow2img(g[case + "_in"].copy())
        assert out.dtype == np.uint8
        np.testing.assert_array_equal(out, g[case + "_out"], err_msg=case)


def test_wrappers_match_reference(golden, oracle_params):
    g = golden("g4_wrappers")
    fr = torch.from_numpy(g["frames"])
    with torch.no_grad():
        d = O.depth_projection(oracle_params, fr, "DepthModule.model.netG.")
        m = O.vos_projection(oracle_params, fr[0], fr[1], "VOSModule.net.")
        big = torch.from_numpy(g["flow_frames"])
        flow = O.flownet2_forward(oracle_params, big.permute(3, 0, 1, 2).unsqueeze(0), "FlowModule.net.")
        pic = O.flow_projection(oracle_params, big[0], big[1], "FlowModule.net.")
    np.testing.assert_allclose(d.numpy(), g["depth"], rtol=0, atol=2e-5)
    np.testing.assert_array_equal(m.numpy(), g["vos_mask"])
    np.testing.assert_allclose(flow.numpy(), g["flow"], rtol=0, atol=1e-5 * np.abs(g["flow"]).max())
    assert (pic.numpy() != g["flow_pic"]).mean() < 1e-3


def test_full_forward_two_recurrent_frames(golden, oracle_params):
    """G6: VSR.forward twice (None -> recurrent estimate) at LR 66x70 (exercises the x64 centre crop)."""
    g = golden("g6_vsr")
    data = torch.from_numpy(g["data"])
    assert bool(g["high_frames1_matches_out0"])
    hf = torch.zeros(3, 4 * data.shape[1], 4 * data.shape[2], 3)
    with torch.no_grad():
        out0 = O.vsr_forward(oracle_params, data.clone(), None, high_frames=hf)
        assert torch.equal(hf[1], out0[0])
        out1 = O.vsr_forward(oracle_params, data.clone(), out0)
    scale = np.abs(g["out0"]).max()
    np.testing.assert_allclose(out0.numpy(), g["out0"], rtol=0, atol=1e-5 * scale)
    np.testing.assert_allclose(out1.numpy(), g["out1"], rtol=0, atol=1e-5 * scale)
