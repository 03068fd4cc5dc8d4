"""The oracle's restatement of the train=True branch (VSR.loss_calculate, video_super_resolution.py:71-80; loss_function.py)
against what the imported reference returned (tests/golden/g10_loss.npz): two recurrent frames, the second one re-using the
object mask cached by the first (defect D7)."""
import numpy as np
import torch

from oracle import vsr_oracle as O


def test_oracle_loss_matches_the_reference(golden, oracle_params):
    g = golden("g10_loss")
    hr = torch.from_numpy(g["hr"].astype(np.float32))
    target = hr[1:2].clone()
    state = {}
    for k, (out, want) in enumerate(((g["out0"], g["loss0"]), (g["out1"], g["loss1"]))):
        hf = hr.clone()
        hf[1] = torch.from_numpy(out)[0]                      # high_frames[1] = output (:66) before loss_calculate (:67)
        taps = {}
        with torch.no_grad():
            loss = O.loss_calculate(oracle_params, target, hf, state, taps)
        assert abs(float(loss) - float(want)) <= 1e-6 * abs(float(want)), (float(loss), float(want))
        # each of genSR, objSR, genFlow, objFlow (video_super_resolution.py:73-79) on its own: the weighted sum hides the flow
        # terms (0.006 * 26.7 in 16364), which is how a wrong masked fill of the flow variant once passed
        for name, got, ref in zip(("genSR", "objSR", "genFlow", "objFlow"), taps["terms"], g["terms"][k]):
            assert abs(got - ref) <= 1e-6 * abs(ref), (k, name, got, ref)
        # what loss4object returned (loss_function.py:87-101), exactly: uint8-valued, masked entries 0 in BOTH variants
        assert np.array_equal(taps["masked_sr_out"].numpy(), g[f"masked_sr_out{k}"].astype(np.float32))
        assert np.array_equal(taps["masked_flow"].numpy(), g[f"masked_flow{k}"].astype(np.float32))
        if k == 0:
            assert np.array_equal(taps["masked_sr_tgt"].numpy(), g["masked_sr_tgt0"].astype(np.float32))
        m = np.broadcast_to(g["mask"].reshape(g["masked_flow0"].shape[1:]), g["masked_flow0"].shape)
        assert (taps["masked_flow"].numpy()[m] == 0).all()
    assert np.array_equal(state["mask"].numpy(), g["mask"])
    assert 0.05 < g["mask"].mean() < 0.95                      # the fixture exercises both branches of the masking


def test_product_keeps_the_reference_loss_module_names(cpu_vsr):
    names = {n for n, _ in cpu_vsr.named_children()}
    assert names == {"model", "FlowModule", "DepthModule", "VOSModule", "SR_loss", "Flow_loss", "loss4object"}   # :15-21
    sd = cpu_vsr.state_dict()
    for k in ("SR_loss.loss_network.0.weight", "SR_loss.loss_network.28.bias", "Flow_loss.SR_loss.loss_network.17.weight",
              "loss4object.VOS.net.fuse.weight"):
        assert k in sd, k
    assert tuple(sd["SR_loss.loss_network.28.weight"].shape) == (512, 512, 3, 3)
    assert not any(p.requires_grad for p in cpu_vsr.SR_loss.loss_network.parameters())   # loss_function.py:14-15
