"""End-to-end `VSR.forward` and its guidance wrappers on the GPU against the reference's golden vectors.

The trunks of FlowNet2 / depth / OSVOS run on stock PyTorch-ROCm (MIOpen) convolutions whose algorithms differ
from ATen-CPU: tolerance 1e-3 of the value range (the north_star bar).  The flow pictures and the VOS mask are
discrete (uint8 colour steps, a 0.7 threshold), so a tiny fraction of pixels may flip; bounded below.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = 1e-3
# fp16 end-to-end bars (measured on MI355X in round 2; see LAB_NOTES.md section 7)
# measured: PSNR(255) 63.1 / 63.8 dB, PSNR(signal span) 50.3 / 50.2 dB, p99 0.61 / 0.58, median 0.07 grey levels
PSNR255_BAR, PSNR_SIGNAL_BAR, P99_BAR = 58.0, 45.0, 1.2


def _relerr(a, ref):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else a
    return np.abs(a - ref).max() / np.abs(ref).max()


def test_wrappers_match_reference_golden(golden, gpu_vsr):
    g = golden("g4_wrappers")
    fr = torch.from_numpy(g["frames"]).cuda()
    assert _relerr(gpu_vsr.DepthModule(fr), g["depth"]) < TOL
    mask = gpu_vsr.VOSModule(fr[0], fr[1]).cpu().numpy()
    assert (mask != g["vos_mask"]).mean() < 5e-3
    big = torch.from_numpy(g["flow_frames"]).cuda()
    flow = gpu_vsr.FlowModule.net(big.permute(3, 0, 1, 2).unsqueeze(0))
    assert _relerr(flow, g["flow"]) < TOL
    pic = gpu_vsr.FlowModule(big[0], big[1]).cpu().numpy()
    assert pic.shape == g["flow_pic"].shape
    d = np.abs(pic - g["flow_pic"])
    # +-1..2 grey-level steps where a 2e-5 flow difference crosses a rounding boundary; the pixel(s) attaining the
    # maximum radius sit exactly on the `rad <= 1` branch of compute_color (flow_utils.py:52-57, a 25 % jump), so a
    # handful of isolated pixels may differ by more
    assert (d > 0).mean() < 0.02 and (d > 2).mean() < 1e-3


def test_full_forward_two_recurrent_frames(golden, gpu_vsr):
    """End to end against the reference's golden frames.  The reference quantises FlowNet2's flow to uint8 colour
    pictures (flow_utils.py:4-24) and thresholds the OSVOS probability at 0.7: an fp32 rounding difference in a
    trunk (MIOpen vs ATen-CPU summation order, ~2e-5 of the flow range) flips a few of those discrete pixels, and
    every flip perturbs a ~29x29 LR-pixel receptive field of the SR net.  So the end-to-end bar is stated on
    image quality (PSNR, tail of the error distribution); the kernels themselves are held to the 1e-3/2e-5 bars
    with the discrete planes teacher-forced (next test)."""
    g = golden("g6_vsr")
    data = torch.from_numpy(g["data"]).cuda()
    hf = torch.zeros(3, 4 * data.shape[1], 4 * data.shape[2], 3, device="cuda")
    out0, loss = gpu_vsr(data, None, hf, None, train=False)
    assert loss is None and out0.shape == (1, 264, 280, 3) and out0.is_cuda
    assert torch.equal(hf[1], out0[0])  # in-place side effect (video_super_resolution.py:66)
    out1, _ = gpu_vsr(data, None, hf, out0, train=False)
    for out, ref in ((out0, g["out0"]), (out1, g["out1"])):
        err = np.abs(out.cpu().numpy() - ref)
        psnr = 10 * np.log10(255.0 ** 2 / max(float(np.mean(err ** 2)), 1e-20))
        assert psnr > 60.0, psnr                      # north_star: within 0.05 dB means >> 60 dB between the two
        assert np.percentile(err, 50) < 1e-3 * 255.0  # median error below 1e-3 of the pixel range
        assert np.percentile(err, 99) < 1.0           # 99 % of the pixels within one grey level


def test_sr_passes_teacher_forced_from_oracle(gpu_vsr, oracle_params):
    """Both SR passes of one frame with the oracle's own 8-plane inputs (its flow pictures, depth planes, masked
    estimate): isolates the hand-written kernels from the discrete guidance planes.  Bar: 2e-5 of the range."""
    from oracle import vsr_oracle as O
    rs = np.random.RandomState(77)
    data = torch.from_numpy(rs.randint(0, 256, (3, 64, 72, 3)).astype(np.float32))
    taps = {}
    with torch.no_grad():
        ref_out = O.vsr_forward(oracle_params, data, None, taps=taps)
    got1 = gpu_vsr.model(taps["pass1_input"].cuda()).cpu().numpy()
    ref1 = taps["pass1_output"].numpy()
    assert np.abs(got1 - ref1).max() <= 2e-5 * np.abs(ref1).max()
    got2 = gpu_vsr.model(taps["pass2_input"].cuda()).permute(0, 2, 3, 1).cpu().numpy()
    assert np.abs(got2 - ref_out.numpy()).max() <= 2e-5 * np.abs(ref_out.numpy()).max()


def test_cpu_input_raises_and_train_true_needs_its_tensors(gpu_vsr):
    x = torch.zeros(3, 64, 64, 3)
    with pytest.raises(RuntimeError):
        gpu_vsr(x, None, None, None, train=False)
    with pytest.raises(ValueError):
        gpu_vsr(x.cuda(), None, None, None)  # train defaults to True like the reference signature: target / high_frames needed


def _check_loss_terms(model, g, rep, bar):
    """Each of genSR, objSR, genFlow, objFlow (video_super_resolution.py:73-79) against the reference's own value, and the
    masked tensors loss4object handed out (loss_function.py:87-101): masked entries are 0 in BOTH variants.  The weighted sum
    alone hides the flow terms (0.006 * 26.7 in 16364): a wrong fill of the flow variant (63) once passed on it."""
    t = model.last_loss_terms
    for name, got, ref in zip(("genSR", "objSR", "genFlow", "objFlow"), t["terms"], g["terms"][rep]):
        rel = abs(got - ref) / abs(ref)
        print(f"  [{name}] {got:.4f} vs reference {ref:.4f} (rel {rel:.2e})")
        assert rel < bar, (rep, name, got, ref)
    m = model.loss4object.mask
    mf = t["masked_flow"]
    assert float(mf[m.reshape(mf.shape[1:]).expand(mf.shape)].abs().max()) == 0.0          # fill_value=0 (:99)
    assert float(t["masked_sr_out"][m.reshape(t["masked_sr_out"].shape[1:]).unsqueeze(0)].abs().max()) == 0.0
    # frames 0 and 2 of high_frames are the fixture's own uint8 frames: where this run's mask agrees with the reference's, the
    # masked tensors must be the reference's exactly (frame 1 is this run's SR output, compared through the terms above)
    same = torch.from_numpy(g["mask"]).to(m.device) == m
    ref_mf = torch.from_numpy(g[f"masked_flow{rep}"].astype(np.float32)).to(mf.device)
    for k in (0, 2):
        ok = same.reshape(mf.shape[1:])
        assert torch.equal(mf[k][ok], ref_mf[k][ok])
    if rep == 0:
        ref_t = torch.from_numpy(g["masked_sr_tgt0"].astype(np.float32)).to(mf.device)
        ok = same.reshape(ref_t.shape[1:]).unsqueeze(0)
        assert torch.equal(t["masked_sr_tgt"][ok], ref_t[ok])


def test_reference_driver_call_with_loss(golden, gpu_vsr):
    """main.py:196-203 replayed verbatim against the drop-in -- positional call, `train` left at its default (True),
    `real_loss.data` read -- on the inputs of the reference-generated fixture g10 (train=True, two recurrent frames).
    The loss is a sum of MSEs over VGG16 features of frames whose guidance planes are discrete (see above), so its bar is
    relative: 2e-3 of the reference's value; the object mask must agree on all but a sliver of pixels."""
    import copy
    g = golden("g10_loss")
    model = copy.deepcopy(gpu_vsr)
    model.loss4object.reset()
    model.keep_loss_terms = True
    hr = torch.from_numpy(g["hr"])                                   # one dataset item, uint8 [3,H,W,3] -> T = 1 below
    datas = hr.unsqueeze(0)
    from video_super_resolution_amd import driver
    data, target, high_frames = driver.ingest_item(datas.cuda(), 4)  # main.py:187-189 (+ MakeCuda :191-194)
    estimated_image = None
    total_loss = []
    for rep, want in enumerate((g["loss0"], g["loss1"])):            # the fixture calls twice with the same item
        hf_item = high_frames.clone()
        for x, y, high_frame in zip(data, target, hf_item):          # main.py:199
            with torch.no_grad():                                    # :200
                output, real_loss = model(x, y, high_frame, estimated_image)   # :201
                estimated_image = output                             # :202
                total_loss.append(real_loss.data)                    # :203
        assert real_loss.device.type == "cpu" and real_loss.dim() == 0
        rel = abs(float(real_loss) - float(want)) / abs(float(want))
        print(f"[train=True call {rep}] loss {float(real_loss):.3f} vs reference {float(want):.3f} (rel {rel:.2e})")
        assert rel < 2e-3
        _check_loss_terms(model, g, rep, 2e-3)
    assert (model.loss4object.mask.cpu().numpy() != g["mask"]).mean() < 5e-3
    assert float(sum(total_loss) / len(total_loss)) > 0                # main.py:207 forms this mean


def test_headline_fp16_configuration_end_to_end(golden, gpu_vsr_f16):
    """The throughput configuration (MFMA SR stack, float16 BatchNorm-folded trunks, batched guidance) against the
    reference's golden frames: image-quality bar (the discrete guidance planes flip a few pixels, see above)."""
    g = golden("g6_vsr")
    data = torch.from_numpy(g["data"]).cuda()
    out0, loss = gpu_vsr_f16(data, None, None, None, train=False)
    out1, _ = gpu_vsr_f16(data, None, None, out0, train=False)
    assert loss is None
    for i, (out, ref) in enumerate(((out0, g["out0"]), (out1, g["out1"]))):
        err = np.abs(out.cpu().numpy() - ref)
        mse = float(np.mean(err ** 2))
        psnr = 10 * np.log10(255.0 ** 2 / mse)                       # at the pixel peak
        span = float(ref.max() - ref.min())
        psnr_sig = 10 * np.log10(span ** 2 / mse)                    # against the golden frame's own value range
        p50, p99 = np.percentile(err, 50), np.percentile(err, 99)
        print(f"[fp16 e2e frame {i}] PSNR(255) {psnr:.2f} dB, PSNR(signal span {span:.1f}) {psnr_sig:.2f} dB, "
              f"median {p50:.4f}, p99 {p99:.4f}, max {err.max():.3f} grey levels")
        # bars = measured (LAB_NOTES.md section 7) minus a margin for box-to-box differences in the discrete planes
        assert psnr > PSNR255_BAR and psnr_sig > PSNR_SIGNAL_BAR and p99 < P99_BAR, (psnr, psnr_sig, p99)


@pytest.mark.parametrize("precision", ["fp32", "fp16"])
def test_psnr_against_the_target_within_0p05_db_of_the_reference(golden, gpu_vsr, gpu_vsr_f16, precision):
    """north_star: "PSNR within 0.05 dB of the reference".  On the reference-generated fixture g10 (a dataset item `hr`, the
    reference's two recurrent output frames `out0` / `out1`): PSNR(this build's frame, HR target) against PSNR(the reference's
    frame, the same target), peak 255, both frames of the recurrence -- in the parity configuration and in the headline
    (fp16 storage) one.  (The other PSNR tests bound PSNR(build, reference) >= 58-63 dB, which implies this for any reference
    PSNR below ~43 dB; here it is asserted as stated.)"""
    from video_super_resolution_amd import driver
    g = golden("g10_loss")
    model = gpu_vsr if precision == "fp32" else gpu_vsr_f16
    data, target, _ = driver.ingest_item(torch.from_numpy(g["hr"]).unsqueeze(0).cuda(), 4)   # main.py:187-189
    tgt = target[0].float().cpu().numpy().astype(np.float64)            # [1,H,W,3]: the HR middle frame

    def psnr(a):
        return 10 * np.log10(255.0 ** 2 / float(np.mean((a.astype(np.float64) - tgt) ** 2)))

    est = None
    for rep, ref in enumerate((g["out0"], g["out1"])):
        with torch.no_grad():
            est, _ = model(data[0], None, None, est, train=False)
        mine, theirs = psnr(est.float().cpu().numpy()), psnr(ref)
        print(f"[{precision} frame {rep}] PSNR vs HR target: build {mine:.4f} dB, reference {theirs:.4f} dB, diff {mine - theirs:+.4f} dB")
        assert abs(mine - theirs) <= 0.05, (rep, mine, theirs)


def test_lr_frame_planes_beside_the_guidance_trunks(gpu_vsr_f16):
    """VSR.overlap_shared (default on, fp16 configuration): the FeedbackBlock maps of the three LR-frame planes are evaluated on a
    side stream while the guidance trunks of pass 1 run; the recurrent frames must equal those of the serial order (the same
    kernels on the same values: planes are independent up to the fusion MLP), and those of the evaluation without plane sharing."""
    import copy
    m = copy.deepcopy(gpu_vsr_f16)
    assert m.overlap_shared and m.share_planes
    clip = torch.from_numpy(np.random.RandomState(9).randint(0, 256, (5, 66, 70, 3)).astype(np.float32)).cuda()

    def run(overlap, share=True):
        m.overlap_shared, m.share_planes = overlap, share
        est, outs = None, []
        for t in range(3):
            est, _ = m(clip[t:t + 3], None, None, est, train=False)
            outs.append(est.clone())
        torch.cuda.synchronize()
        return outs
    try:
        ref, got, plain = run(False), run(True), run(False, share=False)
    finally:
        m.overlap_shared, m.share_planes = True, True
    for a, b, c in zip(got, ref, plain):
        assert torch.equal(a, b) and torch.equal(a, c)


def test_early_planes_beside_the_guidance_trunks(gpu_vsr_f16):
    """VSR.early_planes (default on, fp16 configuration, x4): plane 7 of pass 1 (the resized previous output) joins the LR-frame planes on
    the side stream beside the pass-1 trunks, plane 7 of pass 2 (the masked pass-1 frame) runs behind OSVOS on its stream beside
    FlowNet2; both SR calls then evaluate head + FeedbackBlock on planes 3-6 only.  Same kernels on the same values: the recurrent
    frames must equal those of the order without it, bit for bit (first call: plane 7 = frame 0; later calls: the estimate)."""
    import copy
    m = copy.deepcopy(gpu_vsr_f16)
    assert m.early_planes and m.overlap_shared and m.share_planes
    clip = torch.from_numpy(np.random.RandomState(19).randint(0, 256, (5, 66, 70, 3)).astype(np.float32)).cuda()

    def run(early):
        m.early_planes = early
        est, outs = None, []
        for t in range(3):
            est, _ = m(clip[t:t + 3], None, None, est, train=False)
            outs.append(est.clone())
        torch.cuda.synchronize()
        return outs
    default = m.early_planes
    try:
        ref = run(0)
        # level 1: plane 7 of both passes; 2: + the depth planes of pass 2 behind the hourglass; 3: + the flow-picture planes of pass 1 behind FlowNet2
        for level in (1, 2, 3):
            for a, b in zip(run(level), ref):
                assert torch.equal(a, b), level
    finally:
        m.early_planes = default


@pytest.mark.parametrize("hw", [(66, 70), (540, 960)])
def test_streaming_mode_is_bit_identical_to_per_window_evaluation(gpu_vsr_f16, hw):
    """VSR.temporal_cache (opt-in) at a small size and at the headline size (where the launchers' size thresholds are live): the trunks
    run on the frames / pairs a window does not share with the previous one, with the launchers told the window's full batch
    (vsr_conv2d_route_batch) -- same kernel, tile width and split-K per layer as the per-window evaluation, so every frame is equal
    to it bit for bit (it was PSNR > 55 dB while the routes followed the batch)."""
    import copy
    m = copy.deepcopy(gpu_vsr_f16)
    h, w = hw
    from scipy.ndimage import gaussian_filter
    base = gaussian_filter(np.random.RandomState(6).uniform(0, 255, size=(h + 16, w + 32, 3)).astype(np.float32), sigma=(2, 2, 0))
    clip = torch.from_numpy(np.stack([np.floor(base[k:k + h, 2 * k:2 * k + w]) for k in range(6)]).astype(np.float32)).cuda()

    def run(cache):
        m.temporal_cache = cache
        m.reset_temporal_cache()
        est, outs = None, []
        for t in range(4):
            est, _ = m(clip[t:t + 3], None, None, est, train=False)
            outs.append(est.clone())
        torch.cuda.synchronize()
        return outs
    try:
        ref, got = run(False), run(True)
    finally:
        m.temporal_cache = False
        m.reset_temporal_cache()
    for t, (a, b) in enumerate(zip(got, ref)):
        assert torch.equal(a, b), (t, float((a - b).abs().max()))


@pytest.mark.parametrize("hw", [(66, 70), (540, 960)])
def test_graph_replay_is_bit_identical_to_eager(gpu_vsr_f16, hw):
    """GraphedVSR (opt-in): the ~900 launches of an inference call on its four streams captured once and replayed as one HIP graph --
    a recurrent clip (first call without, later calls with the previous output: two graphs), `high_frames[1]` written as
    video_super_resolution.py:66 does; every frame equal to the eager call's bit for bit.  Training calls are refused."""
    import copy
    from video_super_resolution_amd import GraphedVSR
    m = copy.deepcopy(gpu_vsr_f16)
    h, w = hw
    clip = torch.from_numpy(np.random.RandomState(8).randint(0, 256, (6, h, w, 3)).astype(np.float32)).cuda()
    S = m.upscale_factor

    def run(call):
        est, outs = None, []
        hf = torch.zeros((3, S * h, S * w, 3), dtype=torch.float32, device="cuda")
        for t in range(4):
            est, loss = call(clip[t:t + 3], None, hf, est, train=False)
            assert loss is None and torch.equal(hf[1], est[0])
            outs.append(est.clone())
        torch.cuda.synchronize()
        return outs
    ref = run(m)
    g = GraphedVSR(m)
    got = run(g)
    assert len(g._graphs) == 2
    for t, (a, b) in enumerate(zip(got, ref)):
        assert torch.equal(a, b), (t, float((a - b).abs().max()))
    got2 = run(g)   # replays only
    assert len(g._graphs) == 2 and all(torch.equal(a, b) for a, b in zip(got2, ref))
    with pytest.raises(ValueError):
        g(clip[:3], torch.zeros((1, S * h, S * w, 3), device="cuda"), None, None, train=True)
    if h < 100:   # a weight update (in place: the kernels read packed copies) must not be answered from the old graph
        with torch.no_grad():
            m.model.conv_out[0].bias.add_(3.0)
        new_ref, _ = m(clip[:3], None, None, None, train=False)
        new_got, _ = g(clip[:3], None, None, None, train=False)
        assert torch.equal(new_got, new_ref) and not torch.equal(new_ref, ref[0]) and len(g._graphs) == 3


def test_graph_replay_in_the_float32_configuration(gpu_vsr):
    """GraphedVSR on the exact float32 configuration (own float32 kernels + the stock operators its trunks still route to, all on the
    capturing streams): replayed frames equal the eager ones bit for bit."""
    import copy
    from video_super_resolution_amd import GraphedVSR
    m = copy.deepcopy(gpu_vsr)
    clip = torch.from_numpy(np.random.RandomState(9).randint(0, 256, (5, 66, 70, 3)).astype(np.float32)).cuda()
    est, ref = None, []
    for t in range(3):
        est, _ = m(clip[t:t + 3], None, None, est, train=False)
        ref.append(est.clone())
    g, est = GraphedVSR(m), None
    for t in range(3):
        est, _ = g(clip[t:t + 3], None, None, est, train=False)
        assert torch.equal(est, ref[t]), t


def test_streaming_mode_matches_per_window_evaluation(gpu_vsr_f16):
    """VSR.temporal_cache (opt-in): depth predictions / flow pictures of the two frames consecutive windows share are kept
    across calls (bit-identity with the per-window evaluation: the test above).  Here: cache bookkeeping -- an in-place change of a
    frame must be seen (version counter), not served from the cache; fresh windows never hit."""
    import copy
    m = copy.deepcopy(gpu_vsr_f16)
    clip = torch.from_numpy(np.random.RandomState(5).randint(0, 256, (6, 66, 70, 3)).astype(np.float32)).cuda()

    def run(cache, clip_):
        m.temporal_cache = cache
        m.reset_temporal_cache()
        est, outs = None, []
        for t in range(4):
            est, _ = m(clip_[t:t + 3], None, None, est, train=False)
            outs.append(est.clone())
        return outs
    ref = run(False, clip)
    got = run(True, clip)
    assert torch.equal(got[0], ref[0])                      # nothing cached yet in the first window
    for t, (a, b) in enumerate(zip(got, ref)):
        mse = ((a - b) ** 2).mean().item()
        psnr = 10 * np.log10(255.0 ** 2 / max(mse, 1e-20))
        print(f"[streaming window {t}] PSNR vs per-window evaluation {psnr:.2f} dB")
        assert torch.equal(a, b), (t, psnr)
    assert len(m._tcache["depth"]) == 3 and len(m._tcache["flow"]) == 2
    # a frame changed in place between calls: same storage, new version -> recomputed, not served from the cache
    m.temporal_cache = True
    m.reset_temporal_cache()
    est, _ = m(clip[0:3], None, None, None, train=False)
    hits_before = dict(m._tcache["depth"])
    clip[2].add_(7.0).clamp_(0, 255)
    key_f2 = m._tkey(clip[2])
    assert key_f2 not in hits_before                        # the stale prediction of the old frame contents cannot match
    est2, _ = m(clip[1:4], None, None, est, train=False)
    assert torch.isfinite(est2).all()
    # two DIFFERENT windows built afresh (torch.stack: version 0 every time) back to back.  The caching allocator
    # hands the freed window's address to the next one; a cache that keyed on (address, shape, version) alone then served the
    # first window's depth / flow for the second.  Entries now hold their frames, so the address cannot come back.
    m.reset_temporal_cache()
    rs = np.random.RandomState(11)
    wins = [torch.from_numpy(rs.randint(0, 256, (3, 66, 70, 3)).astype(np.float32)) for _ in range(2)]
    got2 = []
    for wnd in wins:
        fresh = torch.stack([f.cuda() for f in wnd])         # fresh storage, _version == 0
        got2.append(m(fresh, None, None, None, train=False)[0].clone())
        del fresh
    m.temporal_cache = False
    want2 = [m(torch.stack([f.cuda() for f in wnd]), None, None, None, train=False)[0] for wnd in wins]
    assert torch.equal(got2[0], want2[0])
    assert torch.equal(got2[1], want2[1])                    # a stale hit would show the first window's guidance here


def test_reference_driver_call_with_loss_fp16(golden, gpu_vsr_f16):
    """The same replay in the fp16 configuration: SR net on the MFMA path, and the loss's twelve VGG16 passes and its OSVOS on
    the MFMA convolution (trunk_exec.VGGFeatExec / OSVOSExec).  fp16 frames and fp16 feature maps against the reference's
    float32 loss: 2e-3 relative."""
    import copy
    from video_super_resolution_amd import driver
    g = golden("g10_loss")
    model = copy.deepcopy(gpu_vsr_f16)
    model.loss4object.reset()
    model.keep_loss_terms = True
    data, target, high_frames = driver.ingest_item(torch.from_numpy(g["hr"]).unsqueeze(0).cuda(), 4)
    estimated_image = None
    for rep, want in enumerate((g["loss0"], g["loss1"])):
        hf_item = high_frames.clone()
        for x, y, high_frame in zip(data, target, hf_item):
            with torch.no_grad():
                output, real_loss = model(x, y, high_frame, estimated_image)
                estimated_image = output
        rel = abs(float(real_loss) - float(want)) / abs(float(want))
        print(f"[train=True fp16 call {rep}] loss {float(real_loss):.3f} vs reference {float(want):.3f} (rel {rel:.2e})")
        assert rel < 2e-3                                          # measured 1.0e-4 / 6.3e-5
        _check_loss_terms(model, g, rep, 4e-3)
    assert model.SR_loss._exec is not None and model.loss4object._exec is not None   # the MFMA executors ran
    assert (model.loss4object.mask.cpu().numpy() != g["mask"]).mean() < 1e-2
