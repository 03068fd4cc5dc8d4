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


@pytest.mark.parametrize("shape", [(2, 64, 96), (1, 70, 90), (3, 37, 45), (1, 135, 241), (1, 270, 480), (2, 271, 249)])
def test_hourglass_fused_front(gpu_vsr, shape):
    """igemm.HHourglassFront (csrc/conv_hg_front.hip: stem + max pool + the skip inception's fused 1x1s in one launch, the
    128-channel stem map never written) against the three launches it replaces: the whole trunk's prediction and the front's
    three tensors one by one (ragged tiles, odd sizes: floor-mode pooling).  From 65,536 pixels up the three launches are
    k_stem7_rows / k_conv1x1_stream / k_pool2 -- the same MFMA sequences per output as the fused kernel, so the same BITS;
    below that the generic gather kernel serves stem and 1x1 (bias added after the sum instead of first): fp16 rounding apart."""
    from video_super_resolution_amd import igemm
    netg = gpu_vsr.DepthModule.model.netG
    N, h, w = shape
    exact = N * h * w >= 65536
    fr = torch.from_numpy(np.random.RandomState(h + w).randint(0, 256, (N, h, w, 3)).astype(np.float32)).cuda()
    ex = HourglassExec(netg)
    assert ex.front is not None and ex.fused_front

    def same(a, b, what, bar=2e-3):
        if exact:
            assert torch.equal(a, b), what
        else:
            d = (a.float() - b.float()).abs().max().item() / max(b.float().abs().max().item(), 1e-6)
            assert d <= bar, (what, d)

    # the front alone, with the stem's own map written too
    x4 = torch.zeros((N, h, w, 4), dtype=torch.float16, device="cuda")
    x4[..., :3] = fr
    inc = ex.prog[1][1][1][0][1][1][1][0][1]
    stem = ex.prog[1][0][1]
    buf = torch.full((N, h, w, inc.width), 5.0, dtype=torch.float16, device="cuda")
    pooled = torch.empty((N, h // 2, w // 2, 128), dtype=torch.float16, device="cuda")
    smap = torch.empty((N, h, w, 128), dtype=torch.float16, device="cuda")
    ex.front(x4, buf, pooled, smap)
    ref_s = stem(x4)
    same(smap, ref_s[..., :128], "stem map")
    assert torch.equal(pooled, igemm.pool2x2(smap, 0, 128, 0))          # the max pool of the kernel's own map: exact at every size
    ref_b = torch.full_like(buf, 5.0)
    inc.first(smap, out=ref_b, out_coff=0, in_coff=0)
    c2 = inc.first.cout
    same(buf[..., :c2], ref_b[..., :c2], "fused 1x1s")
    assert float((buf[..., c2:] - 5.0).abs().max()) == 0.0
    with torch.no_grad():
        fused = ex(fr).clone()
        ex.fused_front = False
        try:
            three = ex(fr).clone()
        finally:
            ex.fused_front = True
    same(fused, three, "trunk prediction", bar=1e-2)


def test_hourglass_level_streams_without_the_fused_front(gpu_vsr):
    """with the hourglass's per-level side streams on (VSR_HOURGLASS_STREAMS=1, `concurrent`) and the fused front off, the
    level-1 arm ends in a SegMap (dense 16-channel branch maps): `_fan_out` records it on the consumer's stream segment by segment.
    Same launches in another stream order: identical prediction."""
    netg = gpu_vsr.DepthModule.model.netG
    fr = torch.from_numpy(np.random.RandomState(9).randint(0, 256, (2, 70, 90, 3)).astype(np.float32)).cuda()
    ex = HourglassExec(netg)
    with torch.no_grad():
        ex.fused_front = False
        try:
            serial = ex(fr).clone()
            ex.concurrent = True
            try:
                side = ex(fr).clone()
            finally:
                ex.concurrent = False
        finally:
            ex.fused_front = True
    torch.cuda.synchronize()
    assert torch.equal(serial, side)


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


def test_flownet2_fused_heads_against_the_per_head_launches(gpu_vsr):
    """The fused flow heads (igemm.HFlowHead: predict_flow + the next level's flow upsampling in one launch, default) against the
    first build (generic convolution + transposed convolution per head): the same fp16 operands and fp32 sums in another order,
    so the sub-networks' flows agree to fp16 rounding; every head of every sub-network is routed to the fused kernel."""
    from video_super_resolution_amd import _lib as L
    from video_super_resolution_amd import trunk_exec
    net = gpu_vsr.FlowModule.net
    x = torch.from_numpy(np.random.RandomState(5).randint(0, 256, (2, 3, 2, 128, 192)).astype(np.float32)).cuda()
    ex = FlowNet2Exec(net)
    outs = {}
    old = trunk_exec._Refine.fused_heads
    try:
        for fused in (True, False):
            trunk_exec._Refine.fused_heads = fused
            L.ROUTES.calls, L.ROUTES.enabled = [], True
            with torch.no_grad():
                outs[fused] = ex(x).clone()
            L.ROUTES.enabled = False
            heads = [lab for lab, _ in L.ROUTES.calls if lab.startswith("flow_head")]
            assert len(heads) == (4 * 5 + 1 if fused else 0), heads       # C, S1, S2, SD: 5 levels each; fusion: predict_flow2
    finally:
        trunk_exec._Refine.fused_heads = old
        L.ROUTES.enabled = False
    a, b = outs[True], outs[False]
    rng = b.abs().max().item()
    print(f"[fused heads vs per-head launches] max {(a - b).abs().max().item() / rng:.3e} mean {(a - b).abs().mean().item() / rng:.3e} of range")
    assert (a - b).abs().max().item() < 2e-2 * rng and (a - b).abs().mean().item() < 2e-3 * rng


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
# errors recorded in LAB_NOTES.md section 7 plus a margin; every test prints what it measured.
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


def test_flownet2_exec_error_by_sub_network(golden, gpu_vsr):
    """Where the fp16 FlowNet2 executor's error comes from (2.1e-2 of range at the worst pixel of the
    random-pixel golden case, the loosest bar of the suite): every sub-network's executor is TEACHER-FORCED with exactly the
    input its float32 master saw (hooks on flownetc / flownets_1 / flownets_2 / flownets_d / flownetfusion) and its flow is
    compared with the master's, so each line is that sub-network's own fp16 error, not what it inherited.
    Measured: every sub-network is within 1e-3 of its flow range on its own (C 6.0e-4, S1 7.5e-4, S2 6.5e-4, SD 7.9e-4, Fusion
    9.8e-4); the cascade's 2.1e-2 is amplification by the DATA: the golden frames are white noise, so a 1e-3 flow difference
    moves the bilinear warp of the next sub-network's input by whole grey levels.  On a smooth scene the cascade as a whole is
    at 1.2e-3 (tests/test_gpu_trunk_full_size.py)."""
    from video_super_resolution_amd import igemm
    g = golden("g4_wrappers")
    net = gpu_vsr.FlowModule.net
    big = torch.from_numpy(g["flow_frames"]).cuda()
    x = big.permute(3, 0, 1, 2).unsqueeze(0).contiguous()        # [1,3,2,64,128]
    caps = {}
    hooks = [getattr(net, n).register_forward_hook(lambda m, i, o, n=n: caps.__setitem__(n, (i[0].detach().float(), (o[0] if isinstance(o, tuple) else o).detach().float())))
             for n in ("flownetc", "flownets_1", "flownets_2", "flownets_d", "flownetfusion")]
    with torch.no_grad():
        ref = net(x)
        ex = FlowNet2Exec(net)
        whole = ex(x)
    for h_ in hooks:
        h_.remove()
    subs = {"flownetc": (ex.c, 32), "flownets_1": (ex.s1, 16), "flownets_2": (ex.s2, 16), "flownets_d": (ex.sd, 32), "flownetfusion": (ex.fusion, 32)}
    print(f"[FlowNet2Exec whole, fp16 vs fp32 master] max {_rel(whole, ref):.3e} of range")
    worst = 0.0
    for name, (sub, cp) in subs.items():
        inp, out = caps[name]
        with torch.no_grad():
            got = igemm.to_nchw_float(sub(igemm.to_nhwc_half(inp.contiguous(), cp)), 2)
        assert got.shape == out.shape, (name, got.shape, out.shape)
        mx = _rel(got, out)
        mean = (got - out).abs().mean().item() / out.abs().max().item()
        print(f"  [{name:14s} teacher-forced] max {mx:.3e} mean {mean:.3e} of its own flow range ({out.abs().max().item():.3f})")
        worst = max(worst, mx)
        assert mx < 3e-3 and mean < 6e-4, (name, mx, mean)   # measured: 6.0e-4 ... 9.8e-4 max, 1.1e-4 ... 2.0e-4 mean
    assert worst > 0.0
