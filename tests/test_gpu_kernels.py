"""GPU tier 2: every HIP kernel behind the C ABI against the oracle (CPU, float64) on seeded inputs.

Tolerances (fp32 path vs fp64 oracle): activations rel-L2 <= 1e-5, scalar losses rel <= 1e-5,
gradients rel-L2 <= 5e-4 per kernel (5e-3 through the whole network)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import load_golden, rel_l2, synth_vec

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a ROCm GPU")
    from preset_gen_vae_amd import _lib, ops as _ops
    _lib.load()   # raises loudly if the HIP library is missing
    return _ops


def dev(t):
    return t.to(device='cuda', dtype=torch.float32).contiguous()


# (Cb, Cs, k, s, p, Hb, Wb, B) — every reference layer shape family incl. odd sizes, output_padding variants, 1x1
CONV_CASES = [
    (1, 8, 5, 2, 2, 257, 347, 2),     # enc1 / dec8 full size
    (8, 16, 4, 2, 2, 129, 174, 2),    # enc2 / dec7 full size
    (16, 32, 4, 2, 2, 65, 88, 2),     # enc3 / dec6
    (32, 64, 4, 2, 2, 33, 45, 3),     # enc4 / dec5
    (64, 128, 4, 2, 2, 17, 23, 3),    # enc5 / dec4
    (128, 256, 4, 2, 2, 9, 12, 3),    # enc6 / dec3
    (256, 512, 4, 2, 2, 5, 7, 4),     # enc7 / dec2
    (512, 2048, 1, 1, 0, 3, 4, 4),    # enc8 / dec1 (1x1)
    (3, 5, 4, 2, 2, 10, 13, 2),       # ragged generic
    (5, 3, 5, 2, 2, 12, 9, 1),        # ragged generic k5
    (4, 4, 3, 1, 1, 6, 7, 2),         # stride 1
]


def _conv_inputs(case):
    Cb, Cs, k, s, p, Hb, Wb, B = case
    Hs, Ws = (Hb + 2 * p - k) // s + 1, (Wb + 2 * p - k) // s + 1
    big = synth_vec((B, Cb, Hb, Wb), 0.9137, 0.3) * 1.5
    small = synth_vec((B, Cs, Hs, Ws), 0.7719, 1.1)
    w = synth_vec((Cs, Cb, k, k), 0.6180, 0.7) * (1.0 / np.sqrt(Cb * k * k / (s * s)))
    bias_s = synth_vec((Cs,), 1.37, 0.2) * 0.1
    bias_b = synth_vec((Cb,), 1.73, 0.5) * 0.1
    sc_b, sh_b = 1.0 + 0.2 * synth_vec((Cb,), 2.1, 0.1), 0.3 * synth_vec((Cb,), 2.9, 0.6)
    sc_s, sh_s = 1.0 + 0.2 * synth_vec((Cs,), 3.1, 0.4), 0.3 * synth_vec((Cs,), 3.7, 0.9)
    return big, small, w, bias_s, bias_b, sc_b, sh_b, sc_s, sh_s, Hs, Ws


def _conv_inputs_normal(case):
    """Seeded normal operands (activations with a mean, zero-mean output gradient, He-scaled weights) for the tests that
    compare the ERROR of two kernel families against float64: the sinusoid vectors of _conv_inputs cancel to ~1e-5 of the sum
    of magnitudes over a plane, so an error measured on them (5e-6 .. 3e-5 for a weight gradient, either family) is the
    summation order of the partial sums, not the accuracy of the products."""
    Cb, Cs, k, s, p, Hb, Wb, B = case
    big, small, w, bias_s, bias_b, sc_b, sh_b, sc_s, sh_s, Hs, Ws = _conv_inputs(case)
    gen = torch.Generator().manual_seed(1000 * Cb + B)
    big = (torch.randn(big.shape, generator=gen) + 0.5).to(big.dtype)
    small = torch.randn(small.shape, generator=gen).to(small.dtype)
    w = (torch.randn(w.shape, generator=gen) * (1.0 / np.sqrt(Cb * k * k / (s * s)))).to(w.dtype)
    return big, small, w, bias_s, bias_b, sc_b, sh_b, sc_s, sh_s, Hs, Ws


def _affine(t, sc, sh):
    return t * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)


@pytest.mark.parametrize("B", [1, 40, 300])
def test_end_layer_ring_kernels_all_batch_regimes(ops, B):
    """conv_c1_ring.hip (round 6): the 1 <-> 8 channel 5x5 end layers walk a sample in segments of steps with the rows of
    the next step in flight.  B = 2 (CONV_CASES) gives one step per workgroup - prologue only; here B = 40 (13 segments of 2
    steps), B = 300 (2 segments of 13 steps, 600 units on 512 workgroups: the persistent loop) and B = 1.  The producer's
    BatchNorm affine is folded into weights + a border-aware constant: checked against the un-folded float64 convolution,
    with large shifts so that a wrong border term would show."""
    case = (1, 8, 5, 2, 2, 257, 347, B)
    Cb, Cs, k, s, p, Hb, Wb, _ = case
    geom = ops.ConvGeom(Cb, Cs, k, s, p, Hb, Wb)
    gen = torch.Generator().manual_seed(77 + B)
    small = torch.randn((B, Cs, geom.Hs, geom.Ws), generator=gen)
    w = torch.randn((Cs, Cb, k, k), generator=gen) * 0.15
    bias_b = torch.tensor([0.05])
    sc_s = 1.0 + 0.3 * torch.randn(Cs, generator=gen)
    sh_s = 0.7 * torch.randn(Cs, generator=gen)
    ref = F.conv_transpose2d(_affine(small, sc_s, sh_s).double(), w.double(), bias_b.double(), stride=s, padding=p)
    assert ref.shape[-2:] == (Hb, Wb)
    got = ops.conv_up(geom, dev(small), dev(w), dev(bias_b), ops.PGV_ACT_NONE, 0.0, in_scale=dev(sc_s), in_shift=dev(sh_s))
    err = (got.double().cpu() - ref).abs()
    assert rel_l2(got, ref) < 2e-6
    # the border rows / columns on their own (where the shift term differs from the interior's)
    for sl in (np.s_[:, :, :2], np.s_[:, :, -2:], np.s_[:, :, :, :2], np.s_[:, :, :, -2:]):
        assert err[sl].max().item() < 2e-5 * ref.abs().max().item()
    got = ops.conv_up(geom, dev(small), dev(w), dev(bias_b), ops.PGV_ACT_HARDTANH, 0.0)     # no affine
    ref = F.hardtanh(F.conv_transpose2d(small.double(), w.double(), bias_b.double(), stride=s, padding=p))
    assert rel_l2(got, ref) < 2e-6
    # the stride-2 convolution of the 1-channel image (enc1 forward; ring of input rows)
    big = torch.randn((B, Cb, Hb, Wb), generator=gen)
    bias_s = 0.1 * torch.randn(Cs, generator=gen)
    ref = F.leaky_relu(F.conv2d(big.double(), w.double(), bias_s.double(), stride=s, padding=p), 0.1)
    got = ops.conv_down(geom, dev(big), dev(w), dev(bias_s), ops.PGV_ACT_LEAKY_RELU, 0.1)
    assert rel_l2(got, ref) < 2e-6
    # ... and as the output layer's input gradient with the lower block's BatchNorm + activation backward, class sums and
    # bias gradient in its epilogue (pgv_bwd_fuse)
    a = torch.randn(ref.shape, generator=gen)
    coef = torch.cat([1.0 + 0.2 * torch.randn(Cs, generator=gen), 0.05 * torch.randn(2 * Cs, generator=gen)])
    gb = torch.zeros(ops.CLS_COPIES * Cs, device='cuda')
    cls = torch.zeros(ops.CLS_COPIES * 4 * Cs, device='cuda')
    g = F.conv2d(big.double(), w.double(), None, stride=s, padding=p)
    ka, kb, kc = (coef[i * Cs:(i + 1) * Cs].double().view(1, -1, 1, 1) for i in range(3))
    t = ka * g + kb * a.double() + kc
    ref = torch.where(a.double() > 0, t, 0.1 * t)
    got = ops.conv_down(geom, dev(big), dev(w), None, ops.PGV_ACT_NONE, 0.0,
                        bwd_fuse=(dev(a), dev(coef), gb, ops.PGV_ACT_LEAKY_RELU, 0.1, cls, ops.CLS_COPIES))
    assert rel_l2(got, ref) < 2e-6
    ref_cls = torch.stack([ref[:, :, r::2, c::2].sum(dim=(0, 2, 3)) for r in (0, 1) for c in (0, 1)], dim=1)   # [C][4]
    got_cls = cls.view(ops.CLS_COPIES, Cs, 4).sum(0).double().cpu()
    assert (got_cls - ref_cls).abs().max().item() < 2e-5 * ref.abs().sum(dim=(0, 2, 3)).max().item()
    got_gb = gb.view(ops.CLS_COPIES, Cs).sum(0).double().cpu()
    assert (got_gb - ref.sum(dim=(0, 2, 3))).abs().max().item() < 2e-5 * ref.abs().sum(dim=(0, 2, 3)).max().item()
    # the output layer with the reconstruction criterion in its forward epilogue (pgv_conv_up_sqerr, round 6): output, the
    # gradient of its pre-activation for an upstream gradient of 1, criterion value, bias gradient and class sums - in float64,
    # and the separate launches (conv_up + sqerr_act_bwd with g_loss = 1) for the by-products' layout
    from preset_gen_vae_amd import _lib
    target = torch.randn((B, 1, Hb, Wb), generator=gen) * 0.7
    scale = 1.0 / target.numel()
    for act, act_ref, dact in ((ops.PGV_ACT_HARDTANH, F.hardtanh, lambda o: ((o > -1) & (o < 1)).double()),
                               (ops.PGV_ACT_LEAKY_RELU, lambda t_: F.leaky_relu(t_, 0.1), lambda o: torch.where(o > 0, 1.0, 0.1).double())):
        o_ref = act_ref(F.conv_transpose2d(_affine(small, sc_s, sh_s).double(), (2.5 * w).double(), bias_b.double(), stride=s, padding=p))
        g_ref = dact(o_ref) * 2.0 * scale * (o_ref - target.double())
        gbias, loss, cls1 = (torch.zeros(n_, device='cuda') for n_ in (1, 1, ops.CLS_COPIES * 4))
        res = ops.conv_up_sq(geom, dev(small), dev(2.5 * w), dev(bias_b), act, 0.1, dev(target), scale, gbias, loss, cls1,
                             in_scale=dev(sc_s), in_shift=dev(sh_s))
        assert res is not None, "the 8 -> 1 channel 257x347 layer has the fused kernel"
        out, g_y = res
        assert rel_l2(out, o_ref) < 2e-6
        # (the gradient from the DEVICE's output: an output within rounding of the clamp sits on either side of the gate)
        g_dev = dact(out.double().cpu()) * 2.0 * scale * (out.double().cpu() - target.double())
        assert rel_l2(g_y, g_dev) < 1e-6
        near = ((o_ref.abs() - 1.0).abs() < 1e-5) if act == ops.PGV_ACT_HARDTANH else (o_ref.abs() < 1e-5)
        assert (g_y.double().cpu() - g_ref)[~near].abs().max().item() <= 1e-5 * g_ref.abs().max().item()
        assert abs(loss.item() - scale * ((o_ref - target.double()) ** 2).sum().item()) <= 1e-5 * abs(loss.item())
        g_abs = g_ref.abs().sum().item()
        assert abs(gbias.item() - g_ref.sum().item()) <= 2e-5 * g_abs
        ref_c = torch.stack([g_ref[:, 0, r::2, c::2].sum() for r in (0, 1) for c in (0, 1)])
        assert (cls1.view(ops.CLS_COPIES, 4).sum(0).double().cpu() - ref_c).abs().max().item() <= 2e-5 * g_abs
        # the separate launches produce the same tensors
        o2 = ops.conv_up(geom, dev(small), dev(2.5 * w), dev(bias_b), act, 0.1, in_scale=dev(sc_s), in_shift=dev(sh_s))
        g2, gb2, l2, c2 = torch.empty_like(o2), torch.zeros(1, device='cuda'), torch.zeros(1, device='cuda'), torch.zeros_like(cls1)
        ops.sqerr_act_bwd(o2, dev(target), torch.ones((), device='cuda'), scale, act, 0.1, g2, gb2, prezeroed=True, loss_acc=l2, cls=c2)
        assert torch.equal(out, o2) and rel_l2(g_y, g2) < 1e-6
    # kernel policies 1 - 3 and bf16 operand mode have no fused kernel: nothing is launched, the caller is told
    _lib.load().pgv_set_kernel_policy(3)
    try:
        assert ops.conv_up_sq(geom, dev(small), dev(w), dev(bias_b), ops.PGV_ACT_HARDTANH, 0.0, dev(target), scale, gbias, loss, cls1) is None
    finally:
        _lib.load().pgv_set_kernel_policy(0)


# kernel policies (pgv_set_kernel_policy): 0 = tuned (wave-specialised, then band kernels), 3 = the same without the
# wave-specialised generation (band kernels at the reference shapes: the no-workspace wgrad fallback and the bf16 path),
# 2 = runtime-stride MFMA kernels, 1 = generic
@pytest.mark.parametrize("policy", [0, 3, 2, 1])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_down_up_wgrad(ops, case, policy):
    from preset_gen_vae_amd import _lib
    Cb, Cs, k, s, p, Hb, Wb, B = case
    big, small, w, bias_s, bias_b, sc_b, sh_b, sc_s, sh_s, Hs, Ws = _conv_inputs(case)
    geom = ops.ConvGeom(Cb, Cs, k, s, p, Hb, Wb)
    assert (geom.Hs, geom.Ws) == (Hs, Ws)
    lib = _lib.load()
    lib.pgv_set_kernel_policy(policy)
    try:
        # down: conv of the lazily-normalised big tensor, LeakyReLU epilogue, fused BN statistics
        ref = F.leaky_relu(F.conv2d(_affine(big, sc_b, sh_b), w, bias_s, stride=s, padding=p), 0.1)
        stats = torch.empty(2 * Cs, device='cuda', dtype=torch.float64)
        got = ops.conv_down(geom, dev(big), dev(w), dev(bias_s), ops.PGV_ACT_LEAKY_RELU, 0.1, in_scale=dev(sc_b),
                            in_shift=dev(sh_b), stats=stats)
        assert rel_l2(got, ref) < 1e-5
        ref_stats = torch.cat([ref.sum(dim=(0, 2, 3)), (ref * ref).sum(dim=(0, 2, 3))])
        assert rel_l2(stats, ref_stats) < 2e-5
        # down without affine / bias / activation (the form used for ConvTranspose2d input gradients)
        ref = F.conv2d(big, w, None, stride=s, padding=p)
        got = ops.conv_down(geom, dev(big), dev(w), None, ops.PGV_ACT_NONE, 0.0)
        # The structured inputs cancel: at the 5x7 shape |ref| is 1/4000 of the convolution of the absolute values, so
        # rel-L2 measures fp32 summation ORDER (4.8e-6 for the sequential generic kernel, 1.6e-5 for the deep-layer
        # kernel's four K groups; scratch/acc_noise.py).  Where 1e-5 is not met the error must be below one unit
        # roundoff (6e-8) of the scale rounding errors are proportional to - measured 4e-9.
        err_abs = (got.double().cpu() - ref).norm() / F.conv2d(big.abs(), w.abs(), None, stride=s, padding=p).norm()
        assert rel_l2(got, ref) < 1e-5 or (rel_l2(got, ref) < 5e-5 and err_abs.item() < 2e-8)
        # up: transposed conv of the lazily-normalised small tensor; output_padding implied by Hb/Wb
        oph, opw = Hb - ((Hs - 1) * s - 2 * p + k), Wb - ((Ws - 1) * s - 2 * p + k)
        ref = F.conv_transpose2d(_affine(small, sc_s, sh_s), w, bias_b, stride=s, padding=p,
                                 output_padding=(oph, opw))
        ref_act = F.leaky_relu(ref, 0.1)
        stats = torch.empty(2 * Cb, device='cuda', dtype=torch.float64)
        got = ops.conv_up(geom, dev(small), dev(w), dev(bias_b), ops.PGV_ACT_LEAKY_RELU, 0.1, in_scale=dev(sc_s),
                          in_shift=dev(sh_s), stats=stats)
        assert rel_l2(got, ref_act) < 1e-5
        ref_stats = torch.cat([ref_act.sum(dim=(0, 2, 3)), (ref_act * ref_act).sum(dim=(0, 2, 3))])
        assert rel_l2(stats, ref_stats) < 2e-5
        got = ops.conv_up(geom, dev(small), dev(w), dev(bias_b), ops.PGV_ACT_HARDTANH, 0.0, in_scale=dev(sc_s),
                          in_shift=dev(sh_s))
        assert rel_l2(got, F.hardtanh(ref)) < 1e-5
        # wgrad with either operand lazily normalised
        bigr = _affine(big, sc_b, sh_b).requires_grad_(False)
        wv = w.clone().requires_grad_(True)
        y = F.conv2d(bigr, wv, None, stride=s, padding=p)
        y.backward(small)
        gw = torch.empty((Cs, Cb, k, k), device='cuda')
        ops.conv_wgrad(geom, dev(big), dev(small), gw, big_scale=dev(sc_b), big_shift=dev(sh_b))
        assert rel_l2(gw, wv.grad) < 5e-5   # fp32 fma chain over B*Hs*Ws pixels (up to 45k) per output
        wv = w.clone().requires_grad_(True)
        y = F.conv2d(big, wv, None, stride=s, padding=p)
        y.backward(_affine(small, sc_s, sh_s))
        ops.conv_wgrad(geom, dev(big), dev(small), gw, small_scale=dev(sc_s), small_shift=dev(sh_s))
        assert rel_l2(gw, wv.grad) < 5e-5   # fp32 fma chain over B*Hs*Ws pixels (up to 45k) per output
    finally:
        lib.pgv_set_kernel_policy(0)


@pytest.mark.parametrize("case", [(8, 16, 4, 2, 2, 129, 174, 40), (16, 32, 4, 2, 2, 65, 88, 64),
                                  (32, 64, 4, 2, 2, 33, 45, 200)])
def test_band_kernels_many_units(ops, case):
    """The persistent band kernels loop over (sample, band) units with the next unit prefetched in registers: batches
    large enough that every workgroup processes several units (and the last ones fewer), against MIOpen fp32."""
    Cb, Cs, k, s, p, Hb, Wb, B = case
    big, small, w, bias_s, bias_b, sc_b, sh_b, sc_s, sh_s, Hs, Ws = [dev(t) if torch.is_tensor(t) else t
                                                                     for t in _conv_inputs(case)]
    geom = ops.ConvGeom(Cb, Cs, k, s, p, Hb, Wb)
    ref = F.leaky_relu(F.conv2d(_affine(big, sc_b, sh_b), w, bias_s, stride=s, padding=p), 0.1)
    stats = torch.empty(2 * Cs, device='cuda', dtype=torch.float64)
    got = ops.conv_down(geom, big, w, bias_s, ops.PGV_ACT_LEAKY_RELU, 0.1, in_scale=sc_b, in_shift=sh_b, stats=stats)
    assert rel_l2(got, ref) < 1e-5
    assert rel_l2(stats, torch.cat([ref.sum(dim=(0, 2, 3)), (ref * ref).sum(dim=(0, 2, 3))])) < 2e-5
    oph, opw = Hb - ((Hs - 1) * s - 2 * p + k), Wb - ((Ws - 1) * s - 2 * p + k)
    ref = F.leaky_relu(F.conv_transpose2d(_affine(small, sc_s, sh_s), w, bias_b, stride=s, padding=p,
                                          output_padding=(oph, opw)), 0.1)
    stats = torch.empty(2 * Cb, device='cuda', dtype=torch.float64)
    got = ops.conv_up(geom, small, w, bias_b, ops.PGV_ACT_LEAKY_RELU, 0.1, in_scale=sc_s, in_shift=sh_s, stats=stats)
    assert rel_l2(got, ref) < 1e-5
    assert rel_l2(stats, torch.cat([ref.sum(dim=(0, 2, 3)), (ref * ref).sum(dim=(0, 2, 3))])) < 2e-5
    wv = w.clone().requires_grad_(True)
    F.conv2d(_affine(big, sc_b, sh_b), wv, None, stride=s, padding=p).backward(small)
    gw = torch.empty((Cs, Cb, k, k), device='cuda')
    ops.conv_wgrad(geom, big, small, gw, big_scale=sc_b, big_shift=sh_b)
    assert rel_l2(gw, wv.grad) < 2e-4   # fp32 chains over up to 1.2 M pixels per output, summed in a different order
    gw2 = torch.empty_like(gw)
    ops.conv_wgrad(geom, big, small, gw2, big_scale=sc_b, big_shift=sh_b)
    assert rel_l2(gw2, gw) < 1e-5       # float atomics: run-to-run differences stay at rounding level


@pytest.mark.parametrize("shape", [(8, 16, 4, 129, 174), (16, 32, 4, 65, 88), (32, 64, 4, 33, 45), (1, 8, 5, 257, 347),
                                   # deep-layer kernels: sample groups of 1 / 4 / 4 (conv), 1 / 2 / 4 (transposed conv), ragged
                                   # last groups, outputs staged through LDS, K groups reduced through LDS
                                   (64, 128, 4, 17, 23), (128, 256, 4, 9, 12), (256, 512, 4, 5, 7)])
@pytest.mark.parametrize("B", [1, 19, 257])
def test_conv_products_odd_batches(ops, shape, B):
    """The persistent kernels (wave-specialised / direct, XCD-aware unit order, deferred stores, per-workgroup partial
    gradients) at batch sizes that leave workgroups with 0, 1 or an odd number of units, against MIOpen fp32."""
    Cb, Cs, k, Hb, Wb = shape
    torch.manual_seed(B)
    g = ops.ConvGeom(Cb, Cs, k, 2, 2, Hb, Wb)
    big, small = torch.randn(B, Cb, Hb, Wb, device='cuda'), torch.randn(B, Cs, g.Hs, g.Ws, device='cuda')
    w = torch.randn(Cs, Cb, k, k, device='cuda') * 0.1
    bs, bb = torch.randn(Cs, device='cuda'), torch.randn(Cb, device='cuda')
    sc_b, sh_b = torch.rand(Cb, device='cuda') + 0.5, torch.randn(Cb, device='cuda') * 0.1
    sc_s, sh_s = torch.rand(Cs, device='cuda') + 0.5, torch.randn(Cs, device='cuda') * 0.1

    def aff(t, sc, sh):
        return t * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)

    st = torch.zeros(2 * Cs, device='cuda', dtype=torch.float64)
    ref = F.leaky_relu(F.conv2d(aff(big, sc_b, sh_b), w, bs, stride=2, padding=2), 0.1)
    got = ops.conv_down(g, big, w, bs, ops.PGV_ACT_LEAKY_RELU, 0.1, in_scale=sc_b, in_shift=sh_b, stats=st)
    assert rel_l2(got, ref) < 1e-5
    assert rel_l2(st, torch.cat([ref.sum((0, 2, 3)), (ref * ref).sum((0, 2, 3))])) < 2e-5
    assert rel_l2(ops.conv_down(g, big, w, None, ops.PGV_ACT_NONE, 0.0), F.conv2d(big, w, None, stride=2, padding=2)) < 1e-5
    oph, opw = Hb - ((g.Hs - 1) * 2 - 4 + k), Wb - ((g.Ws - 1) * 2 - 4 + k)
    ref = F.leaky_relu(F.conv_transpose2d(aff(small, sc_s, sh_s), w, bb, stride=2, padding=2,
                                          output_padding=(oph, opw)), 0.1)
    stb = torch.zeros(2 * Cb, device='cuda', dtype=torch.float64)
    got = ops.conv_up(g, small, w, bb, ops.PGV_ACT_LEAKY_RELU, 0.1, in_scale=sc_s, in_shift=sh_s, stats=stb)
    assert rel_l2(got, ref) < 1e-5
    assert rel_l2(stb, torch.cat([ref.sum((0, 2, 3)), (ref * ref).sum((0, 2, 3))])) < 2e-5
    ref = F.conv_transpose2d(small, w, None, stride=2, padding=2, output_padding=(oph, opw))
    assert rel_l2(ops.conv_up(g, small, w, None, ops.PGV_ACT_NONE, 0.0), ref) < 1e-5
    gw = torch.empty_like(w)
    wv = w.clone().requires_grad_(True)
    F.conv2d(aff(big, sc_b, sh_b), wv, None, stride=2, padding=2).backward(small)
    ops.conv_wgrad(g, big, small, gw, big_scale=sc_b, big_shift=sh_b)
    assert rel_l2(gw, wv.grad) < 5e-5
    wv = w.clone().requires_grad_(True)
    F.conv2d(big, wv, None, stride=2, padding=2).backward(aff(small, sc_s, sh_s))
    ops.conv_wgrad(g, big, small, gw, small_scale=sc_s, small_shift=sh_s)
    assert rel_l2(gw, wv.grad) < 5e-5


def _bf16(t):
    return t.float().bfloat16().double()


def _affine_fma(t, sc, sh):
    """fmaf(x, scale, shift) as the kernels' loaders evaluate it: one rounding to float32."""
    return (t.float().double() * sc.float().double().view(1, -1, 1, 1) + sh.float().double().view(1, -1, 1, 1)).float()


@pytest.mark.parametrize("policy", [0, 3, 2, 1])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_bf16_operand_mode(ops, case, policy):
    """PGV_COMPUTE_BF16 (BASELINE config 2): both operands of every product rounded to bfloat16 (after the lazy
    normalisation), float32 accumulation — against float64 convolutions of the rounded operands.  Products of bf16
    values are exact in float32, so the tolerance is the fp32 accumulation's, not bf16's."""
    from preset_gen_vae_amd import _lib
    Cb, Cs, k, s, p, Hb, Wb, B = case
    big, small, w, bias_s, bias_b, sc_b, sh_b, sc_s, sh_s, Hs, Ws = _conv_inputs(case)
    geom = ops.ConvGeom(Cb, Cs, k, s, p, Hb, Wb)
    lib = _lib.load()
    lib.pgv_set_kernel_policy(policy)
    ops.set_compute_dtype('bf16')
    try:
        assert ops.compute_dtype() == 'bf16'
        bias_s64, bias_b64 = bias_s.float().double(), bias_b.float().double()
        big_n, small_n = _bf16(_affine_fma(big, sc_b, sh_b)), _bf16(_affine_fma(small, sc_s, sh_s))
        ref = F.leaky_relu(F.conv2d(big_n, _bf16(w), bias_s64, stride=s, padding=p), 0.1)
        ref_fp32 = F.leaky_relu(F.conv2d(_affine(big, sc_b, sh_b), w, bias_s, stride=s, padding=p), 0.1)
        got = ops.conv_down(geom, dev(big), dev(w), dev(bias_s), ops.PGV_ACT_LEAKY_RELU, 0.1, in_scale=dev(sc_b),
                            in_shift=dev(sh_b))
        assert rel_l2(got, ref) < 1e-5
        assert rel_l2(got, ref_fp32) > 2e-4          # the mode really rounds (bf16 operand noise ~ 2^-9 / term)
        oph, opw = Hb - ((Hs - 1) * s - 2 * p + k), Wb - ((Ws - 1) * s - 2 * p + k)
        ref = F.conv_transpose2d(small_n, _bf16(w), bias_b64, stride=s, padding=p, output_padding=(oph, opw))
        got = ops.conv_up(geom, dev(small), dev(w), dev(bias_b), ops.PGV_ACT_NONE, 0.0, in_scale=dev(sc_s),
                          in_shift=dev(sh_s))
        assert rel_l2(got, ref) < 1e-5
        # without the lazy normalisation (the input-gradient form)
        got = ops.conv_up(geom, dev(small), dev(w), None, ops.PGV_ACT_NONE, 0.0)
        ref = F.conv_transpose2d(_bf16(small), _bf16(w), None, stride=s, padding=p, output_padding=(oph, opw))
        assert rel_l2(got, ref) < 1e-5
        got = ops.conv_down(geom, dev(big), dev(w), None, ops.PGV_ACT_NONE, 0.0)
        assert rel_l2(got, F.conv2d(_bf16(big), _bf16(w), None, stride=s, padding=p)) < 1e-5
        # weight gradient: both the (normalised) layer input and the output gradient are rounded
        wv = w.double().clone().requires_grad_(True)
        F.conv2d(big_n, wv, None, stride=s, padding=p).backward(_bf16(small))
        gw = torch.empty((Cs, Cb, k, k), device='cuda')
        ops.conv_wgrad(geom, dev(big), dev(small), gw, big_scale=dev(sc_b), big_shift=dev(sh_b))
        assert rel_l2(gw, wv.grad) < 5e-5
        wv = w.double().clone().requires_grad_(True)
        F.conv2d(_bf16(big), wv, None, stride=s, padding=p).backward(small_n)
        ops.conv_wgrad(geom, dev(big), dev(small), gw, small_scale=dev(sc_s), small_shift=dev(sh_s))
        assert rel_l2(gw, wv.grad) < 5e-5
    finally:
        ops.set_compute_dtype('fp32')
        lib.pgv_set_kernel_policy(0)
    assert ops.compute_dtype() == 'fp32'


@pytest.mark.parametrize("case", [(8, 16, 4, 2, 2, 129, 174, 40), (16, 32, 4, 2, 2, 65, 88, 64),
                                  (32, 64, 4, 2, 2, 33, 45, 70)])
def test_band_kernels_bf16_many_units(ops, case):
    """The bf16 MFMA loops of the persistent band kernels over many work units per workgroup (register prefetch
    ping-pong, per-workgroup BN statistics, fused BN-backward projections all keep their fp32 forms)."""
    Cb, Cs, k, s, p, Hb, Wb, B = case
    big, small, w, bias_s, bias_b, sc_b, sh_b, sc_s, sh_s, Hs, Ws = _conv_inputs(case)
    geom = ops.ConvGeom(Cb, Cs, k, s, p, Hb, Wb)
    ops.set_compute_dtype('bf16')
    try:
        big_n, small_n = _bf16(_affine_fma(big, sc_b, sh_b)), _bf16(_affine_fma(small, sc_s, sh_s))
        ref = F.leaky_relu(F.conv2d(big_n, _bf16(w), bias_s.float().double(), stride=s, padding=p), 0.1)
        stats = torch.empty(2 * Cs, device='cuda', dtype=torch.float64)
        got = ops.conv_down(geom, dev(big), dev(w), dev(bias_s), ops.PGV_ACT_LEAKY_RELU, 0.1, in_scale=dev(sc_b),
                            in_shift=dev(sh_b), stats=stats)
        assert rel_l2(got, ref) < 1e-5
        assert rel_l2(stats, torch.cat([ref.sum(dim=(0, 2, 3)), (ref * ref).sum(dim=(0, 2, 3))])) < 2e-5
        oph, opw = Hb - ((Hs - 1) * s - 2 * p + k), Wb - ((Ws - 1) * s - 2 * p + k)
        ref = F.leaky_relu(F.conv_transpose2d(small_n, _bf16(w), bias_b.float().double(), stride=s, padding=p,
                                              output_padding=(oph, opw)), 0.1)
        stats = torch.empty(2 * Cb, device='cuda', dtype=torch.float64)
        got = ops.conv_up(geom, dev(small), dev(w), dev(bias_b), ops.PGV_ACT_LEAKY_RELU, 0.1, in_scale=dev(sc_s),
                          in_shift=dev(sh_s), stats=stats)
        assert rel_l2(got, ref) < 1e-5
        assert rel_l2(stats, torch.cat([ref.sum(dim=(0, 2, 3)), (ref * ref).sum(dim=(0, 2, 3))])) < 2e-5
        wv = w.double().clone().requires_grad_(True)
        F.conv2d(big_n, wv, None, stride=s, padding=p).backward(_bf16(small))
        gw = torch.empty((Cs, Cb, k, k), device='cuda')
        ops.conv_wgrad(geom, dev(big), dev(small), gw, big_scale=dev(sc_b), big_shift=dev(sh_b))
        assert rel_l2(gw, wv.grad) < 5e-5
    finally:
        ops.set_compute_dtype('fp32')


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
@pytest.mark.parametrize("case", [(64, 128, 4, 2, 2, 17, 23, 5), (128, 256, 4, 2, 2, 9, 12, 9), (256, 512, 4, 2, 2, 5, 7, 9),
                                  (512, 2048, 1, 1, 0, 3, 4, 19)])
def test_deep_kernels_many_units(ops, case, mode):
    """The raw-plane implicit-GEMM kernels of the deep layers (conv_deep.hip) over several sample groups, including a
    partial last group, with lazy normalisation, BN statistics and both operand precisions."""
    Cb, Cs, k, s, p, Hb, Wb, B = case
    big, small, w, bias_s, bias_b, sc_b, sh_b, sc_s, sh_s, Hs, Ws = _conv_inputs(case)
    geom = ops.ConvGeom(Cb, Cs, k, s, p, Hb, Wb)
    rnd = _bf16 if mode == 'bf16' else (lambda t: t.float().double())
    ops.set_compute_dtype(mode)
    try:
        big_n, small_n = rnd(_affine_fma(big, sc_b, sh_b)), rnd(_affine_fma(small, sc_s, sh_s))
        ref = F.leaky_relu(F.conv2d(big_n, rnd(w), bias_s.float().double(), stride=s, padding=p), 0.1)
        stats = torch.empty(2 * Cs, device='cuda', dtype=torch.float64)
        got = ops.conv_down(geom, dev(big), dev(w), dev(bias_s), ops.PGV_ACT_LEAKY_RELU, 0.1, in_scale=dev(sc_b),
                            in_shift=dev(sh_b), stats=stats)
        assert rel_l2(got, ref) < 1e-5
        assert rel_l2(stats, torch.cat([ref.sum(dim=(0, 2, 3)), (ref * ref).sum(dim=(0, 2, 3))])) < 2e-5
        oph, opw = Hb - ((Hs - 1) * s - 2 * p + k), Wb - ((Ws - 1) * s - 2 * p + k)
        ref = F.leaky_relu(F.conv_transpose2d(small_n, rnd(w), bias_b.float().double(), stride=s, padding=p,
                                              output_padding=(oph, opw)), 0.1)
        stats = torch.empty(2 * Cb, device='cuda', dtype=torch.float64)
        got = ops.conv_up(geom, dev(small), dev(w), dev(bias_b), ops.PGV_ACT_LEAKY_RELU, 0.1, in_scale=dev(sc_s),
                          in_shift=dev(sh_s), stats=stats)
        assert rel_l2(got, ref) < 1e-5
        assert rel_l2(stats, torch.cat([ref.sum(dim=(0, 2, 3)), (ref * ref).sum(dim=(0, 2, 3))])) < 2e-5
        wv = w.double().clone().requires_grad_(True)
        F.conv2d(big_n, wv, None, stride=s, padding=p).backward(rnd(small))
        gw = torch.empty((Cs, Cb, k, k), device='cuda')
        ops.conv_wgrad(geom, dev(big), dev(small), gw, big_scale=dev(sc_b), big_shift=dev(sh_b))
        assert rel_l2(gw, wv.grad) < 5e-5
        wv = w.double().clone().requires_grad_(True)
        F.conv2d(rnd(big), wv, None, stride=s, padding=p).backward(small_n)
        ops.conv_wgrad(geom, dev(big), dev(small), gw, small_scale=dev(sc_s), small_shift=dev(sh_s))
        assert rel_l2(gw, wv.grad) < 5e-5
    finally:
        ops.set_compute_dtype('fp32')


@pytest.mark.parametrize("case", [(64, 128, 4, 2, 2, 17, 23, 5), (128, 256, 4, 2, 2, 9, 12, 9), (256, 512, 4, 2, 2, 5, 7, 9),
                                  (256, 512, 4, 2, 2, 5, 7, 16), (128, 192, 4, 2, 2, 9, 12, 3),
                                  (512, 2048, 1, 1, 0, 3, 4, 19), (128, 256, 1, 1, 0, 3, 4, 4),
                                  (32, 64, 4, 2, 2, 33, 45, 5), (32, 64, 4, 2, 2, 33, 45, 70), (16, 32, 4, 2, 2, 65, 88, 5),
                                  (16, 32, 4, 2, 2, 65, 88, 40), (8, 16, 4, 2, 2, 129, 174, 3), (8, 16, 4, 2, 2, 129, 174, 20)])
def test_deep_kernels_bf16_native_with_weight_shadow(ops, case):
    """conv_deep_bf16.hip: the bf16-native kernels of the deep layers behind ``pgv_conv_desc.w_shadow`` (bf16 weight shadow
    written by pgv_conv_weight_shadow), several sample groups with a partial last one: against float64 convolutions of the
    rounded operands, and bit-for-bit deterministic.  Forward form (lazy normalisation, bias, activation, BatchNorm
    statistics) and the input-gradient form (plain product) of both directions."""
    Cb, Cs, k, s, p, Hb, Wb, B = case
    big, small, w, bias_s, bias_b, sc_b, sh_b, sc_s, sh_s, Hs, Ws = _conv_inputs(case)
    geom = ops.ConvGeom(Cb, Cs, k, s, p, Hb, Wb)
    assert ops.conv_weight_shadow(geom, dev(w)) is None            # fp32 mode: no shadow
    ops.set_compute_dtype('bf16')
    try:
        sh = ops.conv_weight_shadow(geom, dev(w))
        assert sh is not None and sh.numel() == 4 * w.numel()      # two bf16 layouts
        big_n, small_n = _bf16(_affine_fma(big, sc_b, sh_b)), _bf16(_affine_fma(small, sc_s, sh_s))
        ref = F.leaky_relu(F.conv2d(big_n, _bf16(w), bias_s.float().double(), stride=s, padding=p), 0.1)
        stats = torch.empty(2 * Cs, device='cuda', dtype=torch.float64)
        got = ops.conv_down(geom, dev(big), dev(w), dev(bias_s), ops.PGV_ACT_LEAKY_RELU, 0.1, in_scale=dev(sc_b),
                            in_shift=dev(sh_b), stats=stats, w_shadow=sh)
        assert rel_l2(got, ref) < 1e-5
        assert rel_l2(stats, torch.cat([ref.sum(dim=(0, 2, 3)), (ref * ref).sum(dim=(0, 2, 3))])) < 2e-5
        old = ops.conv_down(geom, dev(big), dev(w), dev(bias_s), ops.PGV_ACT_LEAKY_RELU, 0.1, in_scale=dev(sc_b),
                            in_shift=dev(sh_b))
        assert rel_l2(got, old) < 2e-6                             # same products, another summation order
        again = ops.conv_down(geom, dev(big), dev(w), dev(bias_s), ops.PGV_ACT_LEAKY_RELU, 0.1, in_scale=dev(sc_b),
                              in_shift=dev(sh_b), w_shadow=sh)
        assert torch.equal(got, again)
        got = ops.conv_down(geom, dev(big), dev(w), None, ops.PGV_ACT_NONE, 0.0, w_shadow=sh)
        assert rel_l2(got, F.conv2d(_bf16(big), _bf16(w), None, stride=s, padding=p)) < 1e-5
        oph, opw = Hb - ((Hs - 1) * s - 2 * p + k), Wb - ((Ws - 1) * s - 2 * p + k)
        ref = F.leaky_relu(F.conv_transpose2d(small_n, _bf16(w), bias_b.float().double(), stride=s, padding=p,
                                              output_padding=(oph, opw)), 0.1)
        stats = torch.empty(2 * Cb, device='cuda', dtype=torch.float64)
        got = ops.conv_up(geom, dev(small), dev(w), dev(bias_b), ops.PGV_ACT_LEAKY_RELU, 0.1, in_scale=dev(sc_s),
                          in_shift=dev(sh_s), stats=stats, w_shadow=sh)
        assert rel_l2(got, ref) < 1e-5
        assert rel_l2(stats, torch.cat([ref.sum(dim=(0, 2, 3)), (ref * ref).sum(dim=(0, 2, 3))])) < 2e-5
        again = ops.conv_up(geom, dev(small), dev(w), dev(bias_b), ops.PGV_ACT_LEAKY_RELU, 0.1, in_scale=dev(sc_s),
                            in_shift=dev(sh_s), w_shadow=sh)
        assert torch.equal(got, again)
        got = ops.conv_up(geom, dev(small), dev(w), None, ops.PGV_ACT_NONE, 0.0, w_shadow=sh)
        ref = F.conv_transpose2d(_bf16(small), _bf16(w), None, stride=s, padding=p, output_padding=(oph, opw))
        assert rel_l2(got, ref) < 1e-5
        # PGV_STATS_COPIES: the statistics may land in any of the partial copies
        C8 = ops.CLS_COPIES
        stc = torch.zeros(C8 * 2 * Cs, device='cuda', dtype=torch.float64)
        got = ops.conv_down(geom, dev(big), dev(w), dev(bias_s), ops.PGV_ACT_LEAKY_RELU, 0.1, stats=stc, prezeroed=True,
                            stats_copies=True, w_shadow=sh)
        assert rel_l2(stc.view(C8, -1).sum(0), torch.cat([got.double().sum(dim=(0, 2, 3)),
                                                          (got.double() ** 2).sum(dim=(0, 2, 3))])) < 2e-5
    finally:
        ops.set_compute_dtype('fp32')


@pytest.mark.parametrize("case", [(64, 128, 4, 2, 2, 17, 23, 5), (128, 256, 4, 2, 2, 9, 12, 9), (256, 512, 4, 2, 2, 5, 7, 9),
                                  (256, 512, 4, 2, 2, 5, 7, 16), (128, 192, 4, 2, 2, 9, 12, 3)])
def test_deep_kernels_fp32_products_as_six_bf16_instructions(ops, case):
    """conv_deep_split.hip (PGV_COMPUTE_F32_SPLIT): the deep layers with every fp32 product as six bf16 matrix instructions
    on exact three-way splits of both operands - against float64 at fp32 tolerances and no further from it than the native
    fp32 kernels are, several sample groups with a partial last one, bit-for-bit deterministic.  Forward form (lazy
    normalisation, bias, activation, BatchNorm statistics) and the input-gradient form (plain product)."""
    Cb, Cs, k, s, p, Hb, Wb, B = case
    big, small, w, bias_s, bias_b, sc_b, sh_b, sc_s, sh_s, Hs, Ws = _conv_inputs_normal(case)
    geom = ops.ConvGeom(Cb, Cs, k, s, p, Hb, Wb)
    big_n = _affine_fma(big, sc_b, sh_b).double()
    ref = F.leaky_relu(F.conv2d(big_n, w.double(), bias_s.double(), stride=s, padding=p), 0.1)
    native = ops.conv_down(geom, dev(big), dev(w), dev(bias_s), ops.PGV_ACT_LEAKY_RELU, 0.1, in_scale=dev(sc_b), in_shift=dev(sh_b))
    native_plain = ops.conv_down(geom, dev(big), dev(w), None, ops.PGV_ACT_NONE, 0.0)
    oph, opw = Hb - ((Hs - 1) * s - 2 * p + k), Wb - ((Ws - 1) * s - 2 * p + k)
    native_up = ops.conv_up(geom, dev(small), dev(w), dev(bias_b), ops.PGV_ACT_LEAKY_RELU, 0.1, in_scale=dev(sc_s), in_shift=dev(sh_s))
    native_up_plain = ops.conv_up(geom, dev(small), dev(w), None, ops.PGV_ACT_NONE, 0.0)
    ops.set_fp32_products('bf16x6')
    try:
        sh = ops.conv_weight_shadow(geom, dev(w))
        assert sh is not None and sh.numel() == 12 * w.numel()    # three bf16 planes in two layouts
        # the three planes of the down layout (fragment order) add up to the weight exactly
        nslab = Cb // 8
        planes = sh[:6 * w.numel()].view(torch.bfloat16).view(Cs // 64, nslab, 2, 4, 3, 2, 4, 16, 8).float().sum(4)
        # [mb][slab][half][kh][mt][kq][m][8 channels] -> w[cs][cb][kh][kw]
        back = planes.permute(0, 2, 4, 6, 1, 7, 3, 5).reshape(Cs, Cb, 4, 4)
        assert torch.equal(back, dev(w))
        stats = torch.empty(2 * Cs, device='cuda', dtype=torch.float64)
        got = ops.conv_down(geom, dev(big), dev(w), dev(bias_s), ops.PGV_ACT_LEAKY_RELU, 0.1, in_scale=dev(sc_b),
                            in_shift=dev(sh_b), stats=stats, w_shadow=sh)
        e_split, e_native = rel_l2(got, ref), rel_l2(native, ref)
        assert e_split < 2e-6 and e_split < 1.25 * e_native + 1e-7, (e_split, e_native)
        assert rel_l2(stats, torch.cat([ref.sum(dim=(0, 2, 3)), (ref * ref).sum(dim=(0, 2, 3))])) < 2e-5
        again = ops.conv_down(geom, dev(big), dev(w), dev(bias_s), ops.PGV_ACT_LEAKY_RELU, 0.1, in_scale=dev(sc_b),
                              in_shift=dev(sh_b), w_shadow=sh)
        assert torch.equal(got, again)
        got = ops.conv_down(geom, dev(big), dev(w), None, ops.PGV_ACT_NONE, 0.0, w_shadow=sh)
        refp = F.conv2d(big.double(), w.double(), None, stride=s, padding=p)     # (structured inputs: the sums cancel)
        e_split, e_native = rel_l2(got, refp), rel_l2(native_plain, refp)
        assert e_split < 1e-5 and e_split < 1.25 * e_native + 1e-7, (e_split, e_native)
        C8 = ops.CLS_COPIES
        stc = torch.zeros(C8 * 2 * Cs, device='cuda', dtype=torch.float64)
        got = ops.conv_down(geom, dev(big), dev(w), dev(bias_s), ops.PGV_ACT_LEAKY_RELU, 0.1, stats=stc, prezeroed=True,
                            stats_copies=True, w_shadow=sh)
        assert rel_l2(stc.view(C8, -1).sum(0), torch.cat([got.double().sum(dim=(0, 2, 3)),
                                                          (got.double() ** 2).sum(dim=(0, 2, 3))])) < 2e-5
        # the up layout: [cb block of 32][cs group of 8][phase][M half][plane][kq = 2 th + tw][m][8 channels]
        planes = sh[6 * w.numel():].view(torch.bfloat16).view(Cb // 32, Cs // 8, 2, 2, 2, 3, 2, 2, 16, 8).float().sum(5)
        # -> [mb][g][ph][pw][half][th][tw][m][c]; w[cs = g*8+c][cb = mb*32+half*16+m][kh = ph+2th][kw = pw+2tw]
        back = planes.permute(1, 8, 0, 4, 7, 5, 2, 6, 3).reshape(Cs, Cb, 4, 4)
        assert torch.equal(back, dev(w))
        refu = F.leaky_relu(F.conv_transpose2d(_affine_fma(small, sc_s, sh_s).double(), w.double(), bias_b.double(), stride=s,
                                               padding=p, output_padding=(oph, opw)), 0.1)
        stats = torch.empty(2 * Cb, device='cuda', dtype=torch.float64)
        got = ops.conv_up(geom, dev(small), dev(w), dev(bias_b), ops.PGV_ACT_LEAKY_RELU, 0.1, in_scale=dev(sc_s),
                          in_shift=dev(sh_s), stats=stats, w_shadow=sh)
        e_split, e_native = rel_l2(got, refu), rel_l2(native_up, refu)
        assert e_split < 2e-6 and e_split < 1.25 * e_native + 1e-7, (e_split, e_native)
        assert rel_l2(stats, torch.cat([refu.sum(dim=(0, 2, 3)), (refu * refu).sum(dim=(0, 2, 3))])) < 2e-5
        again = ops.conv_up(geom, dev(small), dev(w), dev(bias_b), ops.PGV_ACT_LEAKY_RELU, 0.1, in_scale=dev(sc_s),
                            in_shift=dev(sh_s), w_shadow=sh)
        assert torch.equal(got, again)
        got = ops.conv_up(geom, dev(small), dev(w), None, ops.PGV_ACT_NONE, 0.0, w_shadow=sh)
        refp = F.conv_transpose2d(small.double(), w.double(), None, stride=s, padding=p, output_padding=(oph, opw))
        e_split, e_native = rel_l2(got, refp), rel_l2(native_up_plain, refp)
        assert e_split < 1e-5 and e_split < 1.25 * e_native + 1e-7, (e_split, e_native)
        # weight gradient (Wgrad8 with three plane images of both operands), lazy normalisation on either side
        for kw_n, bigd, smalld in (({'big_scale': dev(sc_b), 'big_shift': dev(sh_b)}, big_n, small.double()),
                                   ({'small_scale': dev(sc_s), 'small_shift': dev(sh_s)}, big.double(),
                                    _affine_fma(small, sc_s, sh_s).double())):
            wv = w.double().clone().requires_grad_(True)
            F.conv2d(bigd, wv, None, stride=s, padding=p).backward(smalld)
            gw = torch.empty((Cs, Cb, k, k), device='cuda')
            ops.conv_wgrad(geom, dev(big), dev(small), gw, **kw_n)
            gw2 = torch.full((Cs, Cb, k, k), 7.0, device='cuda')
            ops.conv_wgrad(geom, dev(big), dev(small), gw2, **kw_n)
            assert torch.equal(gw, gw2)
            ops.set_fp32_products('native')
            gwn = torch.empty((Cs, Cb, k, k), device='cuda')
            ops.conv_wgrad(geom, dev(big), dev(small), gwn, **kw_n)
            ops.set_fp32_products('bf16x6')
            e_split, e_native = rel_l2(gw, wv.grad), rel_l2(gwn, wv.grad)
            assert e_split < 5e-6 and e_split < 1.25 * e_native + 1e-7, (e_split, e_native)
        acc = torch.zeros((Cs, Cb, k, k), device='cuda')
        ops.conv_wgrad(geom, dev(big), dev(small), acc, prezeroed=True, small_scale=dev(sc_s), small_shift=dev(sh_s))
        assert rel_l2(acc, gw) < 1e-6
    finally:
        ops.set_fp32_products('native')


@pytest.mark.parametrize("case", [(512, 2048, 1, 1, 0, 3, 4, 19), (128, 256, 1, 1, 0, 3, 4, 4)])
def test_k1_layers_fp32_products_as_six_bf16_instructions(ops, case):
    """conv_deep_split.hip, the 1x1 layers on 3x4 planes (enc8 / dec1) in PGV_COMPUTE_F32_SPLIT mode: both directions against
    float64 at fp32 tolerances, no further from it than the native fp32 kernels, deterministic; and their weight gradient
    (k1_wgrad_split_kernel in conv_deep_bf16.hip)."""
    Cb, Cs, k, s, p, Hb, Wb, B = case
    big, small, w, bias_s, bias_b, sc_b, sh_b, sc_s, sh_s, Hs, Ws = _conv_inputs(case)
    geom = ops.ConvGeom(Cb, Cs, k, s, p, Hb, Wb)
    refd = F.leaky_relu(F.conv2d(_affine_fma(big, sc_b, sh_b).double(), w.double(), bias_s.double()), 0.1)
    refu = F.leaky_relu(F.conv_transpose2d(_affine_fma(small, sc_s, sh_s).double(), w.double(), bias_b.double()), 0.1)
    nat_d = ops.conv_down(geom, dev(big), dev(w), dev(bias_s), ops.PGV_ACT_LEAKY_RELU, 0.1, in_scale=dev(sc_b), in_shift=dev(sh_b))
    nat_u = ops.conv_up(geom, dev(small), dev(w), dev(bias_b), ops.PGV_ACT_LEAKY_RELU, 0.1, in_scale=dev(sc_s), in_shift=dev(sh_s))
    ops.set_fp32_products('bf16x6')
    try:
        sh = ops.conv_weight_shadow(geom, dev(w))
        assert sh is not None and sh.numel() == 12 * w.numel()    # three bf16 planes in two layouts
        fw = sh[:6 * w.numel()].view(torch.bfloat16).view(Cs // 16, Cb // 32, 3, 4, 16, 8).float().sum(2)   # [r16][ks][kq][m][c]
        assert torch.equal(fw.permute(0, 3, 1, 2, 4).reshape(Cs, Cb), dev(w).view(Cs, Cb))
        tr = sh[6 * w.numel():].view(torch.bfloat16).view(Cb // 16, Cs // 32, 3, 4, 16, 8).float().sum(2)
        assert torch.equal(tr.permute(0, 3, 1, 2, 4).reshape(Cb, Cs), dev(w).view(Cs, Cb).t())
        stats = torch.empty(2 * Cs, device='cuda', dtype=torch.float64)
        got = ops.conv_down(geom, dev(big), dev(w), dev(bias_s), ops.PGV_ACT_LEAKY_RELU, 0.1, in_scale=dev(sc_b),
                            in_shift=dev(sh_b), stats=stats, w_shadow=sh)
        e_split, e_native = rel_l2(got, refd), rel_l2(nat_d, refd)
        assert e_split < 2e-6 and e_split < 1.25 * e_native + 1e-7, (e_split, e_native)
        assert rel_l2(stats, torch.cat([refd.sum(dim=(0, 2, 3)), (refd * refd).sum(dim=(0, 2, 3))])) < 2e-5
        assert torch.equal(got, ops.conv_down(geom, dev(big), dev(w), dev(bias_s), ops.PGV_ACT_LEAKY_RELU, 0.1,
                                              in_scale=dev(sc_b), in_shift=dev(sh_b), w_shadow=sh))
        stats = torch.empty(2 * Cb, device='cuda', dtype=torch.float64)
        got = ops.conv_up(geom, dev(small), dev(w), dev(bias_b), ops.PGV_ACT_LEAKY_RELU, 0.1, in_scale=dev(sc_s),
                          in_shift=dev(sh_s), stats=stats, w_shadow=sh)
        e_split, e_native = rel_l2(got, refu), rel_l2(nat_u, refu)
        assert e_split < 2e-6 and e_split < 1.25 * e_native + 1e-7, (e_split, e_native)
        assert rel_l2(stats, torch.cat([refu.sum(dim=(0, 2, 3)), (refu * refu).sum(dim=(0, 2, 3))])) < 2e-5
        got = ops.conv_up(geom, dev(small), dev(w), None, ops.PGV_ACT_NONE, 0.0, w_shadow=sh)
        assert rel_l2(got, F.conv_transpose2d(small.double(), w.double(), None)) < 1e-5

        # ---- weight gradient (k1_wgrad_split_kernel: both operands split where a block of 8 samples is committed) on seeded
        # normal operands, either side lazily normalised, partial last block of samples, accumulate into a zeroed gradient
        gen = torch.Generator().manual_seed(B)
        bign = (torch.randn(big.shape, generator=gen) + 0.5).to(big.dtype)
        smalln = torch.randn(small.shape, generator=gen).to(small.dtype)
        for kw_n, bigd, smalld in (({'big_scale': dev(sc_b), 'big_shift': dev(sh_b)}, _affine_fma(bign, sc_b, sh_b).double(), smalln.double()),
                                   ({'small_scale': dev(sc_s), 'small_shift': dev(sh_s)}, bign.double(),
                                    _affine_fma(smalln, sc_s, sh_s).double()),
                                   ({}, bign.double(), smalln.double())):
            wv = w.double().clone().requires_grad_(True)
            F.conv2d(bigd, wv, None).backward(smalld)
            gw = torch.empty((Cs, Cb, 1, 1), device='cuda')
            ops.conv_wgrad(geom, dev(bign), dev(smalln), gw, **kw_n)
            gw2 = torch.full((Cs, Cb, 1, 1), 7.0, device='cuda')
            ops.conv_wgrad(geom, dev(bign), dev(smalln), gw2, **kw_n)
            assert torch.equal(gw, gw2)
            ops.set_fp32_products('native')
            gwn = torch.empty((Cs, Cb, 1, 1), device='cuda')
            ops.conv_wgrad(geom, dev(bign), dev(smalln), gwn, **kw_n)
            ops.set_fp32_products('bf16x6')
            e_split, e_native = rel_l2(gw, wv.grad), rel_l2(gwn, wv.grad)
            assert e_split < 1e-6 and e_split < 1.25 * e_native + 1e-7, (e_split, e_native)
        acc = torch.zeros((Cs, Cb, 1, 1), device='cuda')
        ops.conv_wgrad(geom, dev(bign), dev(smalln), acc, prezeroed=True)
        assert rel_l2(acc, gw) < 1e-6
    finally:
        ops.set_fp32_products('native')


ENVELOPE_CASES = [(16, 32, 4, 2, 2, 65, 88, 3), (64, 128, 4, 2, 2, 17, 23, 3), (512, 2048, 1, 1, 0, 3, 4, 4)]


@pytest.mark.parametrize("case", ENVELOPE_CASES)
def test_six_instruction_products_accuracy_envelope(ops, case):
    """Where 'bf16x6 is fp32 arithmetic' holds, beyond the N(0,1) operands of the other tests: one
    large-plane, one deep and the 1x1 layer, convolution + weight gradient, against float64 and against the native
    instruction's own error.

    * operands spread over 2^-40 .. 2^40 (and 2^-60 .. 2^60) inside one contraction: same error as native;
    * a tenth of the operands at 2^-120, whose residual planes are bfloat16 denormals: same error as native;
    * everything at 2^-100: same error as native;
    * everything at 2^-118 - the edge of the envelope: the third plane of every operand (2^-16 of the value) underflows the
      bfloat16 range, the products keep 16 - 17 significant bits (measured 8.8e-6 on all three layers).  Activations behind
      a BatchNorm and He-scaled weights are 100 binary orders of magnitude away from it; documented in DESIGN.md 2.3;
    * +-Inf / NaN operands: exactly the outputs the float64 reference makes non-finite are non-finite."""
    Cb, Cs, k, s, p, Hb, Wb, B = case
    Hs, Ws = (Hb + 2 * p - k) // s + 1, (Wb + 2 * p - k) // s + 1
    bs, ss, ws = (B, Cb, Hb, Wb), (B, Cs, Hs, Ws), (Cs, Cb, k, k)
    geom = ops.ConvGeom(Cb, Cs, k, s, p, Hb, Wb)
    gen = torch.Generator().manual_seed(5 + Cb)

    def wide(shape, lo, hi):
        e = torch.rand(shape, generator=gen) * (hi - lo) + lo
        return (torch.sign(torch.randn(shape, generator=gen)) * torch.exp2(e)).float()

    def errors(big, small, w):
        ref = F.conv2d(big.double(), w.double(), None, stride=s, padding=p)
        wv = w.double().clone().requires_grad_(True)
        F.conv2d(big.double(), wv, None, stride=s, padding=p).backward(small.double())
        out = {}
        for mode in ('native', 'bf16x6'):
            ops.set_fp32_products(mode)
            d = ops.conv_down(geom, dev(big), dev(w), None, ops.PGV_ACT_NONE, 0.0, w_shadow=ops.conv_weight_shadow(geom, dev(w)))
            gw = torch.empty(ws, device='cuda')
            ops.conv_wgrad(geom, dev(big), dev(small), gw)
            out[mode] = (rel_l2(d, ref), rel_l2(gw, wv.grad))
        return out

    try:
        mixed = torch.randn(bs, generator=gen)
        tiny = torch.rand(bs, generator=gen) < 0.1
        mixed[tiny] *= 2.0 ** -120
        regimes = {
            '2^-40..2^40': (wide(bs, -40, 40), wide(ss, -20, 20), wide(ws, -40, 40)),
            '2^-60..2^60': (wide(bs, -60, 60), wide(ss, -2, 2), wide(ws, -60, 60)),
            'a tenth at 2^-120': (mixed, torch.randn(ss, generator=gen), 0.1 * torch.randn(ws, generator=gen)),
            'all at 2^-100': (torch.randn(bs, generator=gen) * 2.0 ** -100, torch.randn(ss, generator=gen),
                              torch.randn(ws, generator=gen)),
        }
        for name, (big, small, w) in regimes.items():
            e = errors(big, small, w)
            for i, what in enumerate(('convolution', 'weight gradient')):
                assert e['bf16x6'][i] < 1e-6 and e['bf16x6'][i] <= 1.25 * e['native'][i] + 1e-7, (name, what, e)
        e = errors(torch.randn(bs, generator=gen) * 2.0 ** -118, torch.randn(ss, generator=gen), torch.randn(ws, generator=gen))
        assert e['native'][0] < 1e-6 and e['bf16x6'][0] < 2e-5 and e['bf16x6'][1] < 2e-5, e    # the documented edge
        for bad in (float('inf'), float('-inf'), float('nan')):
            big = torch.randn(bs, generator=gen)
            big[1, 3, Hb // 2, Wb // 2] = bad
            w = torch.randn(ws, generator=gen)
            ref_bad = ~torch.isfinite(F.conv2d(big.double(), w.double(), None, stride=s, padding=p))
            assert ref_bad.any()
            for mode in ('native', 'bf16x6'):
                ops.set_fp32_products(mode)
                d = ops.conv_down(geom, dev(big), dev(w), None, ops.PGV_ACT_NONE, 0.0,
                                  w_shadow=ops.conv_weight_shadow(geom, dev(w)))
                assert torch.equal(~torch.isfinite(d).cpu(), ref_bad), (bad, mode)
    finally:
        ops.set_fp32_products('native')


BIG_SPLIT_CASES = [(8, 16, 4, 2, 2, 129, 174, 2), (8, 16, 4, 2, 2, 129, 174, 9), (16, 32, 4, 2, 2, 65, 88, 3),
                   (16, 32, 4, 2, 2, 65, 88, 40), (32, 64, 4, 2, 2, 33, 45, 5), (32, 64, 4, 2, 2, 33, 45, 70)]


@pytest.mark.parametrize("case", BIG_SPLIT_CASES)
def test_big_plane_kernels_fp32_products_as_six_bf16_instructions(ops, case):
    """conv_big_split.hip (PGV_COMPUTE_F32_SPLIT): the three large-plane k4 s2 p2 layers of the headline stack, both
    directions, with every fp32 product as six bf16 matrix instructions on exact three-way splits - against float64 at fp32
    tolerances and within 1.25 x the native fp32 kernels' own error; batches with fewer units than workgroups and with
    several units per workgroup; bit-for-bit repeatable.  Forward form (lazy normalisation, bias, activation, BatchNorm
    statistics, statistics copies) and the fused input-gradient form (pgv_bwd_fuse: bias-gradient copies, class sums)."""
    Cb, Cs, k, s, p, Hb, Wb, B = case
    big, small, w, bias_s, bias_b, sc_b, sh_b, sc_s, sh_s, Hs, Ws = _conv_inputs_normal(case)
    geom = ops.ConvGeom(Cb, Cs, k, s, p, Hb, Wb)
    assert ops.conv_weight_shadow(geom, dev(w)) is None          # native fp32 products: no shadow
    oph, opw = Hb - ((Hs - 1) * s - 2 * p + k), Wb - ((Ws - 1) * s - 2 * p + k)
    refd = F.leaky_relu(F.conv2d(_affine_fma(big, sc_b, sh_b).double(), w.double(), bias_s.double(), stride=s, padding=p), 0.1)
    refu = F.leaky_relu(F.conv_transpose2d(_affine_fma(small, sc_s, sh_s).double(), w.double(), bias_b.double(), stride=s,
                                           padding=p, output_padding=(oph, opw)), 0.1)
    nat_d = ops.conv_down(geom, dev(big), dev(w), dev(bias_s), ops.PGV_ACT_LEAKY_RELU, 0.1, in_scale=dev(sc_b), in_shift=dev(sh_b))
    nat_u = ops.conv_up(geom, dev(small), dev(w), dev(bias_b), ops.PGV_ACT_LEAKY_RELU, 0.1, in_scale=dev(sc_s), in_shift=dev(sh_s))
    C8 = ops.CLS_COPIES
    ops.set_fp32_products('bf16x6')
    try:
        sh = ops.conv_weight_shadow(geom, dev(w))
        assert sh is not None and sh.numel() == 12 * w.numel()    # three bf16 planes, fragment order, both directions
        # down layout [M tile][K step = kh * (Cb/8) + g][plane][kq][m][kw][channel of pair 4g + kq]: the planes add up to
        # the weight exactly
        planes = sh[:6 * w.numel()].view(torch.bfloat16).view(Cs // 16, 4, Cb // 8, 3, 4, 16, 4, 2).float().sum(3)
        back = planes.permute(0, 4, 2, 3, 6, 1, 5).reshape(Cs, Cb, 4, 4)     # [mt][m][g][kq][c][kh][kw]
        assert torch.equal(back, dev(w))
        # up layout [M tile][g][plane][kq = 2 th + tw][m][8 small channels], M rows r = (2 ph + pw) * Cb + cb
        planes = sh[6 * w.numel():].view(torch.bfloat16).view(Cb // 4, Cs // 8, 3, 2, 2, 16, 8).float().sum(2)
        rows = planes.permute(0, 4, 1, 5, 2, 3).reshape(2, 2, Cb, Cs, 2, 2)     # [ph][pw][cb][cs][th][tw]
        back = rows.permute(3, 2, 4, 0, 5, 1).reshape(Cs, Cb, 4, 4)           # kh = 2 th + ph, kw = 2 tw + pw
        assert torch.equal(back, dev(w))

        # ---- convolution: forward form
        stats = torch.empty(2 * Cs, device='cuda', dtype=torch.float64)
        got = ops.conv_down(geom, dev(big), dev(w), dev(bias_s), ops.PGV_ACT_LEAKY_RELU, 0.1, in_scale=dev(sc_b),
                            in_shift=dev(sh_b), stats=stats, w_shadow=sh)
        e_split, e_native = rel_l2(got, refd), rel_l2(nat_d, refd)
        assert e_split < 2e-6 and e_split < 1.25 * e_native + 1e-7, (e_split, e_native)
        assert rel_l2(stats, torch.cat([refd.sum(dim=(0, 2, 3)), (refd * refd).sum(dim=(0, 2, 3))])) < 2e-5
        again = ops.conv_down(geom, dev(big), dev(w), dev(bias_s), ops.PGV_ACT_LEAKY_RELU, 0.1, in_scale=dev(sc_b),
                              in_shift=dev(sh_b), w_shadow=sh)
        assert torch.equal(got, again)
        stc = torch.zeros(C8 * 2 * Cs, device='cuda', dtype=torch.float64)
        got = ops.conv_down(geom, dev(big), dev(w), dev(bias_s), ops.PGV_ACT_LEAKY_RELU, 0.1, stats=stc, prezeroed=True,
                            stats_copies=True, w_shadow=sh)
        assert rel_l2(stc.view(C8, -1).sum(0), torch.cat([got.double().sum(dim=(0, 2, 3)),
                                                          (got.double() ** 2).sum(dim=(0, 2, 3))])) < 2e-5
        # ---- convolution as the input gradient of a transposed convolution: plain and with the fused backward epilogue
        prod = F.conv2d(big.double(), w.double(), None, stride=s, padding=p)     # (structured inputs: the sums cancel)
        got = ops.conv_down(geom, dev(big), dev(w), None, ops.PGV_ACT_NONE, 0.0, w_shadow=sh)
        nat = ops.conv_down(geom, dev(big), dev(w), None, ops.PGV_ACT_NONE, 0.0)
        e_split, e_native = rel_l2(got, prod), rel_l2(nat, prod)
        assert e_split < 1e-5 and e_split < 1.25 * e_native + 1e-7, (e_split, e_native)
        for act in (ops.PGV_ACT_LEAKY_RELU, ops.PGV_ACT_NONE):
            a = (dev(small) * 1.3 + 0.1).contiguous()
            coef = dev(torch.cat([1.0 + 0.3 * synth_vec((Cs,), 4.1, 0.2), 0.05 * synth_vec((Cs,), 4.7, 0.3),
                                  0.02 * synth_vec((Cs,), 5.3, 0.8)]))
            gbc, cls = torch.zeros(C8 * Cs, device='cuda'), torch.zeros(C8 * 4 * Cs, device='cuda')
            out = ops.conv_down(geom, dev(big), dev(w), None, ops.PGV_ACT_NONE, 0.0,
                                bwd_fuse=(a, coef, gbc, act, 0.1, cls, C8), w_shadow=sh)
            refb = _bwd_apply_ref(prod.cuda(), a, coef, act, 0.1)
            assert rel_l2(out, refb) < 2e-6, rel_l2(out, refb)
            l1 = refb.abs().sum(dim=(0, 2, 3))
            assert ((gbc.view(C8, Cs).double().sum(0) - refb.sum(dim=(0, 2, 3))).abs() <= 2e-6 * l1 + 1e-12).all()
            ref_cls = torch.stack([refb[:, :, r::2, c::2].sum(dim=(0, 2, 3)) for r in range(2) for c in range(2)], dim=1)
            assert ((cls.view(C8, Cs, 4).double().sum(0) - ref_cls).abs() <= 2e-6 * l1.view(-1, 1) + 1e-12).all()

        # ---- transposed convolution: forward form
        stats = torch.empty(2 * Cb, device='cuda', dtype=torch.float64)
        got = ops.conv_up(geom, dev(small), dev(w), dev(bias_b), ops.PGV_ACT_LEAKY_RELU, 0.1, in_scale=dev(sc_s),
                          in_shift=dev(sh_s), stats=stats, w_shadow=sh)
        e_split, e_native = rel_l2(got, refu), rel_l2(nat_u, refu)
        assert e_split < 2e-6 and e_split < 1.25 * e_native + 1e-7, (e_split, e_native)
        assert rel_l2(stats, torch.cat([refu.sum(dim=(0, 2, 3)), (refu * refu).sum(dim=(0, 2, 3))])) < 2e-5
        again = ops.conv_up(geom, dev(small), dev(w), dev(bias_b), ops.PGV_ACT_LEAKY_RELU, 0.1, in_scale=dev(sc_s),
                            in_shift=dev(sh_s), w_shadow=sh)
        assert torch.equal(got, again)
        stc = torch.zeros(C8 * 2 * Cb, device='cuda', dtype=torch.float64)
        got = ops.conv_up(geom, dev(small), dev(w), dev(bias_b), ops.PGV_ACT_LEAKY_RELU, 0.1, stats=stc, prezeroed=True,
                          stats_copies=True, w_shadow=sh)
        assert rel_l2(stc.view(C8, -1).sum(0), torch.cat([got.double().sum(dim=(0, 2, 3)),
                                                          (got.double() ** 2).sum(dim=(0, 2, 3))])) < 2e-5
        # ---- ... as the input gradient of a convolution: plain and fused (bias-gradient copies)
        prod = F.conv_transpose2d(small.double(), w.double(), None, stride=s, padding=p, output_padding=(oph, opw))
        got = ops.conv_up(geom, dev(small), dev(w), None, ops.PGV_ACT_NONE, 0.0, w_shadow=sh)
        nat = ops.conv_up(geom, dev(small), dev(w), None, ops.PGV_ACT_NONE, 0.0)
        e_split, e_native = rel_l2(got, prod), rel_l2(nat, prod)
        assert e_split < 1e-5 and e_split < 1.25 * e_native + 1e-7, (e_split, e_native)
        for act in (ops.PGV_ACT_LEAKY_RELU, ops.PGV_ACT_NONE):
            a = (dev(big) * 1.3 + 0.1).contiguous()
            coef = dev(torch.cat([1.0 + 0.3 * synth_vec((Cb,), 4.1, 0.2), 0.05 * synth_vec((Cb,), 4.7, 0.3),
                                  0.02 * synth_vec((Cb,), 5.3, 0.8)]))
            gbc = torch.zeros(C8 * Cb, device='cuda')
            out = ops.conv_up(geom, dev(small), dev(w), None, ops.PGV_ACT_NONE, 0.0,
                              bwd_fuse=(a, coef, gbc, act, 0.1, None, C8), w_shadow=sh)
            refb = _bwd_apply_ref(prod.cuda(), a, coef, act, 0.1)
            assert rel_l2(out, refb) < 2e-6
            l1 = refb.abs().sum(dim=(0, 2, 3))
            assert ((gbc.view(C8, Cb).double().sum(0) - refb.sum(dim=(0, 2, 3))).abs() <= 2e-6 * l1 + 1e-12).all()

        # ---- weight gradient (conv_wgrad_split.hip: both operands split in the loader), lazy normalisation on either side
        big_n = _affine_fma(big, sc_b, sh_b).double()
        for kw_n, bigd, smalld in (({'big_scale': dev(sc_b), 'big_shift': dev(sh_b)}, big_n, small.double()),
                                   ({'small_scale': dev(sc_s), 'small_shift': dev(sh_s)}, big.double(),
                                    _affine_fma(small, sc_s, sh_s).double()),
                                   ({}, big.double(), small.double())):
            wv = w.double().clone().requires_grad_(True)
            F.conv2d(bigd, wv, None, stride=s, padding=p).backward(smalld)
            gw = torch.empty((Cs, Cb, k, k), device='cuda')
            ops.conv_wgrad(geom, dev(big), dev(small), gw, **kw_n)
            gw2 = torch.full((Cs, Cb, k, k), 7.0, device='cuda')
            ops.conv_wgrad(geom, dev(big), dev(small), gw2, **kw_n)
            assert torch.equal(gw, gw2)
            ops.set_fp32_products('native')
            gwn = torch.empty((Cs, Cb, k, k), device='cuda')
            ops.conv_wgrad(geom, dev(big), dev(small), gwn, **kw_n)
            ops.set_fp32_products('bf16x6')
            e_split, e_native = rel_l2(gw, wv.grad), rel_l2(gwn, wv.grad)
            assert e_split < 1e-6 and e_split < 1.25 * e_native + 1e-7, (e_split, e_native)
        acc = torch.zeros((Cs, Cb, k, k), device='cuda')
        ops.conv_wgrad(geom, dev(big), dev(small), acc, prezeroed=True)
        assert rel_l2(acc, gw) < 1e-6
    finally:
        ops.set_fp32_products('native')


@pytest.mark.parametrize("shape", [(256, 512, 3, 4), (19, 256, 5, 7), (256, 256, 5, 7), (37, 128, 9, 12), (2, 96, 5, 7)])
@pytest.mark.parametrize("act", [1, 2, 0])
def test_bn_act_bwd_fused_equals_the_two_passes(ops, shape, act):
    """pgv_bn_act_bwd_fused (one launch, a workgroup per channel holding its values in registers) against
    pgv_bn_bwd_reduce + pgv_act_bn_bwd on the same inputs - g_y, bias gradient, gamma / beta gradients - and against float64;
    in place (g_y = g_o) as the backward pass calls it.  Shapes it does not serve are reported by pgv_bn_act_bwd_fusable."""
    B, C, H, W = shape
    HW = H * W
    assert ops.bn_act_bwd_fusable(B, C, HW)
    assert not ops.bn_act_bwd_fusable(256, 64, 17 * 23) and not ops.bn_act_bwd_fusable(256, 16, 12) and \
        not ops.bn_act_bwd_fusable(256, 8, 129 * 174) and not ops.bn_act_bwd_fusable(256, 128, 9 * 12)
    g = dev(synth_vec((B, C, H, W), 0.731, 0.2) + 0.03)
    a = dev(synth_vec((B, C, H, W), 1.377, 0.9) * 1.5)
    mean = a.mean(dim=(0, 2, 3)).contiguous()
    rstd = (1.0 / torch.sqrt(a.var(dim=(0, 2, 3), unbiased=False) + 1e-5)).contiguous()
    scale = (dev(1.0 + 0.2 * synth_vec((C,), 2.1, 0.4)) * rstd).contiguous()
    slope = 0.1
    # two passes
    red = torch.zeros(2 * C, device='cuda', dtype=torch.float64)
    ops.bn_bwd_reduce(g, a, mean, rstd, red, prezeroed=True)
    gy2, gb2, gg2, gbt2 = torch.empty_like(g), torch.zeros(C, device='cuda'), torch.empty(C, device='cuda'), torch.empty(C, device='cuda')
    ops.act_bn_bwd(g, a, scale, mean, rstd, red, act, slope, gy2, gb2, ggamma=gg2, gbeta=gbt2, prezeroed=True)
    # fused, in place
    gy1, gb1, gg1, gbt1 = g.clone(), torch.zeros(C, device='cuda'), torch.empty(C, device='cuda'), torch.empty(C, device='cuda')
    ops.bn_act_bwd_fused(gy1, a, scale, mean, rstd, act, slope, gy1, gb1, ggamma=gg1, gbeta=gbt1, prezeroed=True)
    # float64 reference
    gd, ad = g.double(), a.double()
    ah = (ad - mean.double().view(1, -1, 1, 1)) * rstd.double().view(1, -1, 1, 1)
    n = B * HW
    s0, d0 = gd.sum(dim=(0, 2, 3)), (gd * ah).sum(dim=(0, 2, 3))
    ga = scale.double().view(1, -1, 1, 1) * (gd - s0.view(1, -1, 1, 1) / n - ah * d0.view(1, -1, 1, 1) / n)
    if act == 1:
        ref = torch.where(ad > 0, ga, slope * ga)
    elif act == 2:
        ref = torch.where((ad > -1) & (ad < 1), ga, torch.zeros_like(ga))
    else:
        ref = ga
    assert rel_l2(gy1, ref) < 2e-6 and rel_l2(gy2, ref) < 2e-6
    assert rel_l2(gy1, gy2) < 5e-7
    l1 = ref.abs().sum(dim=(0, 2, 3))
    assert ((gb1.double() - ref.sum(dim=(0, 2, 3))).abs() <= 2e-6 * l1 + 1e-9).all()
    # (the projections are sums of fp32 products with cancellation: both forms sit a few 1e-6 from float64, and on each other)
    assert rel_l2(gg1, d0) < 5e-6 and rel_l2(gbt1, s0) < 5e-6 and rel_l2(gg2, d0) < 5e-6
    assert rel_l2(gg1, gg2) < 1e-6 and rel_l2(gbt1, gbt2) < 1e-6


@pytest.mark.parametrize("shape", [(256, 2048, 3, 4), (19, 2048, 3, 4), (7, 256, 4, 4), (3, 300, 5, 7)])
@pytest.mark.parametrize("act", [1, 2, 0])
def test_act_bwd_without_batchnorm_small_planes(ops, shape, act):
    """pgv_act_bn_bwd with scale = NULL (a block without BatchNorm: enc8 of the 8-layer stack) - small planes whose size is a
    multiple of 4 take the flat walk (act_bwd_flat_kernel), the others the channel walk: g_y = act'(a) g, bias gradient, in
    place, and accumulation into a pre-zeroed bias gradient."""
    B, C, H, W = shape
    g = dev(synth_vec((B, C, H, W), 0.731, 0.2) + 0.03)
    a = dev(synth_vec((B, C, H, W), 1.377, 0.9) * 1.5)
    ref = g.double()
    if act == 1:
        ref = torch.where(a > 0, ref, 0.1 * ref)
    elif act == 2:
        ref = torch.where((a > -1) & (a < 1), ref, torch.zeros_like(ref))
    gy, gb = g.clone(), torch.full((C,), float('nan'), device='cuda')
    ops.act_bn_bwd(gy, a, None, None, None, None, act, 0.1, gy, gb)          # in place, gbias cleared by the call
    assert rel_l2(gy, ref) < 1e-7 and torch.equal(gy == 0, ref == 0)
    l1 = ref.abs().sum(dim=(0, 2, 3))
    assert ((gb.double() - ref.sum(dim=(0, 2, 3))).abs() <= 2e-6 * l1 + 1e-9).all()
    gb2 = torch.ones(C, device='cuda')
    ops.act_bn_bwd(g, a, None, None, None, None, act, 0.1, torch.empty_like(g), gb2, prezeroed=True)   # accumulates
    assert ((gb2.double() - 1.0 - ref.sum(dim=(0, 2, 3))).abs() <= 2e-6 * (l1 + 1.0) + 1e-9).all()


def test_weight_shadows_of_a_stack_in_one_launch(ops):
    """pgv_conv_weight_shadows (one launch for the layers of a conv stack) writes exactly what pgv_conv_weight_shadow writes
    layer by layer; layers without a shadow come back as None; fp32 mode has none at all."""
    shapes = [(8, 16, 4, 2, 2, 129, 174), (16, 32, 4, 2, 2, 65, 88), (32, 64, 4, 2, 2, 33, 45), (64, 128, 4, 2, 2, 17, 23),
              (128, 256, 4, 2, 2, 9, 12), (256, 512, 4, 2, 2, 5, 7), (512, 2048, 1, 1, 0, 3, 4)]
    pairs = []
    for i, (Cb, Cs, k, s, p, Hb, Wb) in enumerate(shapes):
        w = dev(synth_vec((Cs, Cb, k, k), 0.37 + i, 0.11 * i) * 0.05)
        pairs.append((ops.ConvGeom(Cb, Cs, k, s, p, Hb, Wb), w))
    assert ops.conv_weight_shadows(pairs) == [None] * len(pairs)
    ops.set_compute_dtype('bf16')
    try:
        pairs.insert(0, (ops.ConvGeom(1, 8, 5, 2, 2, 257, 347), dev(synth_vec((8, 1, 5, 5), 0.2, 0.3))))
        many = ops.conv_weight_shadows(pairs)
        assert many[0] is None and all(m is not None for m in many[1:])
        for (g, w), m in zip(pairs[1:], many[1:]):
            one = ops.conv_weight_shadow(g, w)
            assert m.numel() == one.numel() == 4 * w.numel() and torch.equal(m, one)
            assert m.data_ptr() % 16 == 0
    finally:
        ops.set_compute_dtype('fp32')


@pytest.mark.parametrize("case", [(8, 16, 4, 2, 2, 129, 174, 2), (1, 8, 5, 2, 2, 257, 347, 2), (64, 128, 4, 2, 2, 17, 23, 3),
                                  (3, 5, 4, 2, 2, 10, 13, 2)])
def test_conv_prezeroed_outputs_accumulate(ops, case):
    """PGV_PREZEROED (pgv_conv_desc.flags): stats / gw are accumulated into, not cleared - every kernel family."""
    Cb, Cs, k, s, p, Hb, Wb, B = case
    big, small, w, bias_s, bias_b, sc_b, sh_b, sc_s, sh_s, Hs, Ws = [dev(t) if torch.is_tensor(t) else t
                                                                     for t in _conv_inputs(case)]
    geom = ops.ConvGeom(Cb, Cs, k, s, p, Hb, Wb)
    st1 = torch.empty(2 * Cs, device='cuda', dtype=torch.float64)
    ops.conv_down(geom, big, w, bias_s, ops.PGV_ACT_LEAKY_RELU, 0.1, stats=st1)
    st2 = st1.clone()
    ops.conv_down(geom, big, w, bias_s, ops.PGV_ACT_LEAKY_RELU, 0.1, stats=st2, prezeroed=True)
    assert rel_l2(st2, 2 * st1) < 1e-9
    st1 = torch.empty(2 * Cb, device='cuda', dtype=torch.float64)
    ops.conv_up(geom, small, w, bias_b, ops.PGV_ACT_LEAKY_RELU, 0.1, stats=st1)
    st2 = st1.clone()
    ops.conv_up(geom, small, w, bias_b, ops.PGV_ACT_LEAKY_RELU, 0.1, stats=st2, prezeroed=True)
    assert rel_l2(st2, 2 * st1) < 1e-9
    gw1 = torch.empty((Cs, Cb, k, k), device='cuda')
    ops.conv_wgrad(geom, big, small, gw1)
    gw2 = gw1.clone()
    ops.conv_wgrad(geom, big, small, gw2, prezeroed=True)
    assert rel_l2(gw2, 2 * gw1) < 1e-5


BWD_FUSE_CASES = [(8, 16, 4, 2, 2, 129, 174, 40), (16, 32, 4, 2, 2, 65, 88, 64), (32, 64, 4, 2, 2, 33, 45, 6),
                  (64, 128, 4, 2, 2, 17, 23, 3), (3, 5, 4, 2, 2, 10, 13, 2), (1, 8, 5, 2, 2, 257, 347, 21),
                  (128, 256, 4, 2, 2, 9, 12, 7), (256, 512, 4, 2, 2, 5, 7, 9), (512, 2048, 1, 1, 0, 3, 4, 19),
                  (1, 8, 5, 2, 2, 257, 347, 70)]   # > 1024 units: persistent direct kernel, 2 units per WG


def _bwd_apply_ref(g, a, coef, act, slope):
    C = a.shape[1]
    ka, kb, kc = [coef[i * C:(i + 1) * C].double().view(1, -1, 1, 1) for i in range(3)]
    t = g.double() * ka + a.double() * kb + kc
    if act == 1:
        t = torch.where(a > 0, t, slope * t)
    elif act == 2:
        t = torch.where((a > -1) & (a < 1), t, torch.zeros_like(t))
    return t


@pytest.mark.parametrize("policy", [0, 3, 2, 1])
@pytest.mark.parametrize("case", BWD_FUSE_CASES)
def test_conv_bwd_fuse_matches_separate_pass(ops, case, policy):
    """pgv_bwd_fuse: an input-gradient call that applies the lower block's BatchNorm + activation backward in its
    epilogue (fused band / wave-specialised / direct epilogues, and the in-place pass every other kernel family falls
    back to) equals the plain product followed by act'(a) * (ka*g + kb*a + kc) in float64; the bias gradient is the
    per-channel sum of the result.  All kernel policies, all three activations."""
    from preset_gen_vae_amd import _lib
    Cb, Cs, k, s, p, Hb, Wb, B = case
    big, small, w, bias_s, bias_b, sc_b, sh_b, sc_s, sh_s, Hs, Ws = [dev(t) if torch.is_tensor(t) else t
                                                                     for t in _conv_inputs(case)]
    geom = ops.ConvGeom(Cb, Cs, k, s, p, Hb, Wb)
    lib = _lib.load()
    lib.pgv_set_kernel_policy(policy)
    try:
        for out_is_big in (True, False):
            C = Cb if out_is_big else Cs
            a = ((big if out_is_big else small) * 1.3 + 0.1).contiguous()    # "saved activation" of the block below
            coef = dev(torch.cat([1.0 + 0.3 * synth_vec((C,), 4.1, 0.2), 0.05 * synth_vec((C,), 4.7, 0.3),
                                  0.02 * synth_vec((C,), 5.3, 0.8)]))
            for act, slope in ((ops.PGV_ACT_LEAKY_RELU, 0.1), (ops.PGV_ACT_HARDTANH, 0.0), (ops.PGV_ACT_NONE, 0.0)):
                gb = torch.zeros(C, device='cuda')
                if out_is_big:
                    out = ops.conv_up(geom, small, w, None, ops.PGV_ACT_NONE, 0.0, bwd_fuse=(a, coef, gb, act, slope))
                    g = ops.conv_up(geom, small, w, None, ops.PGV_ACT_NONE, 0.0)
                else:
                    out = ops.conv_down(geom, big, w, None, ops.PGV_ACT_NONE, 0.0, bwd_fuse=(a, coef, gb, act, slope))
                    g = ops.conv_down(geom, big, w, None, ops.PGV_ACT_NONE, 0.0)
                ref = _bwd_apply_ref(g, a, coef, act, slope)
                assert rel_l2(out, ref) < 2e-6, (out_is_big, act, rel_l2(out, ref))
                l1 = ref.abs().sum(dim=(0, 2, 3))
                err = (gb.double() - ref.sum(dim=(0, 2, 3))).abs()
                assert (err <= 2e-6 * l1 + 1e-12).all(), (out_is_big, act, (err / l1).max().item())
                # class sums of the result as a by-product (kept in the epilogue where the kernel can, else a pass)
                gb3, cls = torch.zeros(C, device='cuda'), torch.zeros(ops.CLS_COPIES * 4 * C, device='cuda')
                fz = (a, coef, gb3, act, slope, cls)
                out3 = (ops.conv_up(geom, small, w, None, ops.PGV_ACT_NONE, 0.0, bwd_fuse=fz) if out_is_big else
                        ops.conv_down(geom, big, w, None, ops.PGV_ACT_NONE, 0.0, bwd_fuse=fz))
                assert rel_l2(out3, ref) < 2e-6
                ref_cls = torch.stack([ref[:, :, r::2, c::2].sum(dim=(0, 2, 3)) for r in range(2) for c in range(2)], dim=1)
                # (kept as partial copies, one per XCD of the producing workgroups; the consumers add them up)
                assert ((cls.double().view(ops.CLS_COPIES, C, 4).sum(0) - ref_cls).abs() <= 2e-6 * l1.view(C, 1) + 1e-12).all()
                assert ((gb3.double() - ref.sum(dim=(0, 2, 3))).abs() <= 2e-6 * l1 + 1e-12).all()
                # the unfused entry point gives the same (in place)
                gb2 = torch.empty(C, device='cuda')
                g2 = g.clone()
                ops.act_bwd_coef(g2, a, coef, act, slope, g2, gb2)
                assert rel_l2(g2, ref) < 2e-6
                assert ((gb2.double() - ref.sum(dim=(0, 2, 3))).abs() <= 2e-6 * l1 + 1e-12).all()
    finally:
        lib.pgv_set_kernel_policy(0)


@pytest.mark.parametrize("case", [(8, 16, 4, 2, 2, 129, 174, 40), (16, 32, 4, 2, 2, 65, 88, 9), (32, 64, 4, 2, 2, 33, 45, 6)])
def test_conv_bwd_fuse_bf16_operand_mode(ops, case):
    """The fused backward epilogue in bf16 operand mode (PGV_COMPUTE_BF16): the 129x174 input gradient runs the
    wave-specialised kernel with the operands rounded in the lean loader's commit, 33x45 the rolling-window form, 65x88 the
    band kernel's bf16 loop.  The epilogue input must be the bf16-mode product itself (float64 convolution of the rounded
    operands, 1e-5), and the fused result the separate pass applied to it."""
    Cb, Cs, k, s, p, Hb, Wb, B = case
    big, small, w, *_ = _conv_inputs(case)
    geom = ops.ConvGeom(Cb, Cs, k, s, p, Hb, Wb)
    Hs, Ws = geom.Hs, geom.Ws
    oph, opw = Hb - ((Hs - 1) * s - 2 * p + k), Wb - ((Ws - 1) * s - 2 * p + k)
    prod_up = F.conv_transpose2d(_bf16(small), _bf16(w), None, stride=s, padding=p, output_padding=(oph, opw))
    prod_down = F.conv2d(_bf16(big), _bf16(w), None, stride=s, padding=p)
    big, small, w = dev(big), dev(small), dev(w)
    ops.set_compute_dtype('bf16')
    try:
        for out_is_big, prod in ((True, prod_up), (False, prod_down)):
            C = Cb if out_is_big else Cs
            a = ((big if out_is_big else small) * 1.3 + 0.1).contiguous()
            coef = dev(torch.cat([1.0 + 0.3 * synth_vec((C,), 4.1, 0.2), 0.05 * synth_vec((C,), 4.7, 0.3),
                                  0.02 * synth_vec((C,), 5.3, 0.8)]))
            gb = torch.zeros(C, device='cuda')
            fz = (a, coef, gb, ops.PGV_ACT_LEAKY_RELU, 0.1)
            out = (ops.conv_up(geom, small, w, None, ops.PGV_ACT_NONE, 0.0, bwd_fuse=fz) if out_is_big else
                   ops.conv_down(geom, big, w, None, ops.PGV_ACT_NONE, 0.0, bwd_fuse=fz))
            ref = _bwd_apply_ref(prod.cuda(), a, coef, ops.PGV_ACT_LEAKY_RELU, 0.1)
            assert rel_l2(out, ref) < 1e-5, (out_is_big, rel_l2(out, ref))
            l1 = ref.abs().sum(dim=(0, 2, 3))
            assert ((gb.double() - ref.sum(dim=(0, 2, 3))).abs() <= 1e-5 * l1 + 1e-12).all()
            # with the bf16 weight shadow (the 33x45 transposed product then runs conv_deep_bf16.hip's persistent kernel),
            # the bias gradient kept as per-XCD partial copies as in the train step
            sh = ops.conv_weight_shadow(geom, w)
            if sh is not None:
                gbc = torch.zeros(ops.CLS_COPIES * C, device='cuda')
                fzc = (a, coef, gbc, ops.PGV_ACT_LEAKY_RELU, 0.1, None, ops.CLS_COPIES)
                out2 = (ops.conv_up(geom, small, w, None, ops.PGV_ACT_NONE, 0.0, bwd_fuse=fzc, w_shadow=sh) if out_is_big else
                        ops.conv_down(geom, big, w, None, ops.PGV_ACT_NONE, 0.0, bwd_fuse=fzc, w_shadow=sh))
                assert rel_l2(out2, ref) < 1e-5, (out_is_big, rel_l2(out2, ref))
                gsum = gbc.view(ops.CLS_COPIES, C).double().sum(0)
                assert ((gsum - ref.sum(dim=(0, 2, 3))).abs() <= 1e-5 * l1 + 1e-12).all()
    finally:
        ops.set_compute_dtype('fp32')


@pytest.mark.parametrize("case", CONV_CASES + [(8, 16, 4, 2, 2, 129, 174, 24), (16, 32, 4, 2, 2, 64, 87, 5),
                                               (1, 8, 5, 2, 2, 257, 347, 9), (2, 3, 2, 2, 0, 8, 10, 3)])
def test_conv_tap_sums(ops, case):
    """pgv_conv_tap_sums = the weight gradient a channel of ones in the OTHER tensor would receive (autograd of
    conv2d / conv_transpose2d over a ones input), both directions."""
    Cb, Cs, k, s, p, Hb, Wb, B = case
    geom = ops.ConvGeom(Cb, Cs, k, s, p, Hb, Wb)
    Hs, Ws = geom.Hs, geom.Ws
    gy_s = synth_vec((B, Cs, Hs, Ws), 0.7719, 1.1) + 0.05
    gy_b = synth_vec((B, Cb, Hb, Wb), 0.9137, 0.3) + 0.05
    # gy small: T[cs][kh][kw] = d/dw <conv2d(ones_big, w[Cs,1,k,k]), gy>
    wv = torch.zeros((Cs, 1, k, k), dtype=torch.float64, requires_grad=True)
    F.conv2d(torch.ones((B, 1, Hb, Wb), dtype=torch.float64), wv, None, stride=s, padding=p).backward(gy_s)
    T = ops.conv_tap_sums(geom, dev(gy_s), False)
    l1 = gy_s.abs().sum().item() / Cs
    assert (T.cpu().view(Cs, k, k) - wv.grad[:, 0]).abs().max().item() <= 1e-6 * l1
    # border form: class sums (here the plain channel sums) minus the positions a tap cannot pair
    cls = ops.conv_class_sums(geom, dev(gy_s), False)
    assert (cls.cpu().double() - gy_s.sum(dim=(0, 2, 3))).abs().max().item() <= 1e-6 * l1
    Tb = ops.conv_tap_sums(geom, dev(gy_s), False, cls=cls)
    assert (Tb.cpu().view(Cs, k, k) - wv.grad[:, 0]).abs().max().item() <= 2e-6 * l1
    # gy big: T[cb][kh][kw] = d/dw <conv_transpose2d(ones_small, w[1,Cb,k,k]), gy>
    oph, opw = Hb - ((Hs - 1) * s - 2 * p + k), Wb - ((Ws - 1) * s - 2 * p + k)
    wv = torch.zeros((1, Cb, k, k), dtype=torch.float64, requires_grad=True)
    F.conv_transpose2d(torch.ones((B, 1, Hs, Ws), dtype=torch.float64), wv, None, stride=s, padding=p,
                       output_padding=(oph, opw)).backward(gy_b)
    T = torch.zeros(Cb * k * k, device='cuda', dtype=torch.float64)
    ops.conv_tap_sums(geom, dev(gy_b), True, T, prezeroed=True)
    l1 = gy_b.abs().sum().item() / Cb
    assert (T.cpu().view(Cb, k, k) - wv.grad[0]).abs().max().item() <= 1e-6 * l1
    if s <= 3:
        cls = ops.conv_class_sums(geom, dev(gy_b), True)
        ref_cls = torch.stack([gy_b[:, :, r::s, c::s].sum(dim=(0, 2, 3)) for r in range(s) for c in range(s)], dim=1)
        assert (cls.cpu().double().view(ops.CLS_COPIES, Cb, s * s).sum(0) - ref_cls).abs().max().item() <= 1e-6 * l1
        Tb = ops.conv_tap_sums(geom, dev(gy_b), True, cls=cls)
        assert (Tb.cpu().view(Cb, k, k) - wv.grad[0]).abs().max().item() <= 2e-6 * l1


@pytest.mark.parametrize("policy", [0, 3])
@pytest.mark.parametrize("case", [(8, 16, 4, 2, 2, 129, 174, 3), (16, 32, 4, 2, 2, 65, 88, 3), (32, 64, 4, 2, 2, 33, 45, 5),
                                  (1, 8, 5, 2, 2, 257, 347, 2), (3, 5, 4, 2, 2, 10, 13, 2)])
def test_bias_gradient_through_partial_copies(ops, case, policy):
    """pgv_bwd_fuse.gbias_copies + pgv_bias_req: a fused input gradient adds the lower block's bias gradient into per-XCD
    partial copies, the reduce launch of that block's own weight gradient adds them up - the same bias gradient as the
    one-copy form, the same weight gradient as without the request (also on top of an earlier value: accumulate)."""
    from preset_gen_vae_amd import _lib
    Cb, Cs, k, s, p, Hb, Wb, B = case
    geom = ops.ConvGeom(Cb, Cs, k, s, p, Hb, Wb)
    lib = _lib.load()
    lib.pgv_set_kernel_policy(policy)
    try:
        for up in (False, True):   # consumer is a Conv2d (its input gradient = conv_up) / a ConvTranspose2d (conv_down)
            gy_shape = (B, Cs, geom.Hs, geom.Ws) if not up else (B, Cb, Hb, Wb)
            lo_shape = (B, Cb, Hb, Wb) if not up else (B, Cs, geom.Hs, geom.Ws)
            Cl = lo_shape[1]
            gy, a = dev(synth_vec(gy_shape, 0.7719, 1.1)), dev(synth_vec(lo_shape, 0.9137, 0.3) * 1.5)
            w = dev(synth_vec((Cs, Cb, k, k), 0.6180, 0.7) * 0.1)
            coef = torch.cat([torch.ones(Cl, device='cuda'), 0.01 * dev(synth_vec((2 * Cl,), 1.7, 0.2))])
            conv = ops.conv_up if not up else ops.conv_down
            gb1 = torch.zeros(Cl, device='cuda')
            g1 = conv(geom, gy, w, None, 0, 0.0, bwd_fuse=(a, coef, gb1, 1, 0.1))
            copies = torch.zeros(ops.CLS_COPIES * Cl, device='cuda')
            g2 = conv(geom, gy, w, None, 0, 0.0, bwd_fuse=(a, coef, copies, 1, 0.1, None, ops.CLS_COPIES))
            assert torch.equal(g1, g2)
            # the lower block's own weight gradient (any geometry whose gradient tensor is g2 will do: reuse this one with
            # g2 in the role of the tensor of its shape) carries the bias request
            big_t, small_t = (g2, dev(synth_vec((B, Cs, geom.Hs, geom.Ws), 0.31, 0.4))) if not up else \
                (dev(synth_vec((B, Cb, Hb, Wb), 0.31, 0.4)), g2)
            gw_ref, gw = torch.empty_like(w), torch.empty_like(w)
            ops.conv_wgrad(geom, big_t, small_t, gw_ref)
            for acc, start in ((False, float('nan')), (True, 0.25)):
                gb2 = torch.full((Cl,), start, device='cuda')
                ops.conv_wgrad(geom, big_t, small_t, gw, bias_finish=(copies, gb2, acc))
                # (not bit-equal: the round-1 weight-gradient kernels accumulate with float atomics)
                assert (gw - gw_ref).abs().max().item() <= 1e-5 * gw_ref.abs().max().item()
                tol = 2e-6 * g1.abs().sum().item() / Cl + 1e-6
                assert (gb2 - (0.25 if acc else 0.0) - gb1).abs().max().item() <= tol, (up, acc)
    finally:
        lib.pgv_set_kernel_policy(0)


@pytest.mark.parametrize("case", [(8, 16, 4, 2, 2, 129, 174, 3), (16, 32, 4, 2, 2, 65, 88, 3), (32, 64, 4, 2, 2, 33, 45, 5),
                                  (64, 128, 4, 2, 2, 17, 23, 19), (128, 256, 4, 2, 2, 9, 12, 19), (256, 512, 4, 2, 2, 5, 7, 21),
                                  (512, 2048, 1, 1, 0, 3, 4, 19)])
def test_bf16_native_kernels_finalize_the_input_batchnorm(ops, case):
    """The bf16-native kernels behind the weight shadow take ``in_bn`` too (pgv_bn_src: the producer's BatchNorm finalized in
    the consumer's prologue): bit for bit the output, vectors and running statistics of pgv_bn_finalize + the plain call."""
    Cb, Cs, k, s, p, Hb, Wb, B = case
    geom = ops.ConvGeom(Cb, Cs, k, s, p, Hb, Wb)
    ops.set_compute_dtype('bf16')
    try:
        for up in (False, True):
            C, H, W = (Cs, geom.Hs, geom.Ws) if up else (Cb, Hb, Wb)
            Co = Cb if up else Cs
            x = dev(synth_vec((B, C, H, W), 0.371, 0.2) * 1.3 + 0.1)
            w = dev(synth_vec((Cs, Cb, k, k), 0.6180, 0.7) * (1.0 / np.sqrt(C * k * k)))
            sh = ops.conv_weight_shadow(geom, w)
            assert sh is not None
            bias = dev(synth_vec((Co,), 1.1, 0.3) * 0.1)
            gamma, beta = dev(1.0 + 0.3 * synth_vec((C,), 2.1, 0.1)), dev(0.3 * synth_vec((C,), 2.9, 0.6))
            stats = torch.zeros(2 * C, device='cuda', dtype=torch.float64)
            ops.bn_stats(x, stats)
            fn = ops.conv_up if up else ops.conv_down
            res = []
            for fused in (False, True):
                rm, rv = dev(synth_vec((C,), 0.5, 0.5)), dev(synth_vec((C,), 0.7, 0.1).abs() + 0.5)
                nbt = torch.tensor(3, device='cuda', dtype=torch.int64)
                vec = [torch.full((C,), float('nan'), device='cuda') for _ in range(4)]
                src = ops.bn_src(stats, B * H * W, gamma, beta, 1e-5, 0.1, rm, rv, nbt, *vec)
                if fused:
                    y = fn(geom, x, w, bias, ops.PGV_ACT_LEAKY_RELU, 0.1, in_bn=src, w_shadow=sh)
                else:
                    ops.bn_src_finalize(src)
                    y = fn(geom, x, w, bias, ops.PGV_ACT_LEAKY_RELU, 0.1, in_scale=vec[0], in_shift=vec[1], w_shadow=sh)
                res.append([y, rm, rv, nbt] + vec)
            for a, b in zip(*res):
                assert torch.equal(a, b)
            assert res[1][3].item() == 4
    finally:
        ops.set_compute_dtype('fp32')


@pytest.mark.parametrize("case", [(64, 128, 4, 2, 2, 17, 23, 19), (128, 256, 4, 2, 2, 9, 12, 19), (256, 512, 4, 2, 2, 5, 7, 21),
                                  (512, 2048, 1, 1, 0, 3, 4, 19), (8, 16, 4, 2, 2, 129, 174, 9), (16, 32, 4, 2, 2, 65, 88, 19),
                                  (32, 64, 4, 2, 2, 33, 45, 31)])
def test_split_product_kernels_finalize_the_input_batchnorm(ops, case):
    """The PGV_COMPUTE_F32_SPLIT kernels take ``in_bn`` as well (the 1x1 ones too, unlike their native fp32 counterparts): bit
    for bit the output, vectors and running statistics of pgv_bn_finalize + the plain call, ragged sample groups."""
    Cb, Cs, k, s, p, Hb, Wb, B = case
    geom = ops.ConvGeom(Cb, Cs, k, s, p, Hb, Wb)
    ops.set_fp32_products('bf16x6')
    try:
        for up in (False, True):
            C, H, W = (Cs, geom.Hs, geom.Ws) if up else (Cb, Hb, Wb)
            Co = Cb if up else Cs
            x = dev(synth_vec((B, C, H, W), 0.371, 0.2) * 1.3 + 0.1)
            w = dev(synth_vec((Cs, Cb, k, k), 0.6180, 0.7) * (1.0 / np.sqrt(C * k * k)))
            sh = ops.conv_weight_shadow(geom, w)
            assert sh is not None
            bias = dev(synth_vec((Co,), 1.1, 0.3) * 0.1)
            gamma, beta = dev(1.0 + 0.3 * synth_vec((C,), 2.1, 0.1)), dev(0.3 * synth_vec((C,), 2.9, 0.6))
            stats = torch.zeros(2 * C, device='cuda', dtype=torch.float64)
            ops.bn_stats(x, stats)
            fn = ops.conv_up if up else ops.conv_down
            res = []
            for fused in (False, True):
                rm, rv = dev(synth_vec((C,), 0.5, 0.5)), dev(synth_vec((C,), 0.7, 0.1).abs() + 0.5)
                nbt = torch.tensor(3, device='cuda', dtype=torch.int64)
                vec = [torch.full((C,), float('nan'), device='cuda') for _ in range(4)]
                src = ops.bn_src(stats, B * H * W, gamma, beta, 1e-5, 0.1, rm, rv, nbt, *vec)
                if fused:
                    y = fn(geom, x, w, bias, ops.PGV_ACT_LEAKY_RELU, 0.1, in_bn=src, w_shadow=sh)
                else:
                    ops.bn_src_finalize(src)
                    y = fn(geom, x, w, bias, ops.PGV_ACT_LEAKY_RELU, 0.1, in_scale=vec[0], in_shift=vec[1], w_shadow=sh)
                res.append([y, rm, rv, nbt] + vec)
            for a, b in zip(*res):
                assert torch.equal(a, b)
            assert res[1][3].item() == 4
            # and without a shadow the call computes natively under the flag
            y_native = fn(geom, x, w, bias, ops.PGV_ACT_LEAKY_RELU, 0.1, in_scale=res[0][4], in_shift=res[0][5])
            assert rel_l2(y_native, res[0][0]) < 1e-5      # (structured inputs: the sums cancel)
    finally:
        ops.set_fp32_products('native')


@pytest.mark.parametrize("policy", [0, 3])
@pytest.mark.parametrize("case", [(8, 16, 4, 2, 2, 129, 174, 3), (16, 32, 4, 2, 2, 65, 88, 3), (32, 64, 4, 2, 2, 33, 45, 5),
                                  (1, 8, 5, 2, 2, 257, 347, 2), (64, 128, 4, 2, 2, 17, 23, 3), (3, 5, 4, 2, 2, 10, 13, 2),
                                  # deep-layer kernels (statistics copies per XCD since ABI v9), ragged sample groups
                                  (64, 128, 4, 2, 2, 17, 23, 19), (128, 256, 4, 2, 2, 9, 12, 19),
                                  (256, 512, 4, 2, 2, 5, 7, 21), (512, 2048, 1, 1, 0, 3, 4, 19)])
def test_conv_finalizes_the_input_batchnorm(ops, case, policy):
    """pgv_conv_down_bn / pgv_conv_up_bn / pgv_dropout_fwd_bn (pgv_bn_src): the consumer kernel finalizes its input's
    BatchNorm in its prologue - the same output, the same scale / shift / mean / rstd vectors, the same running
    statistics and counter as pgv_bn_finalize followed by the plain call (bit for bit: the same float64 expressions)."""
    from preset_gen_vae_amd import _lib
    from preset_gen_vae_amd.rng import DeviceRNG
    Cb, Cs, k, s, p, Hb, Wb, B = case
    geom = ops.ConvGeom(Cb, Cs, k, s, p, Hb, Wb)
    lib = _lib.load()
    lib.pgv_set_kernel_policy(policy)
    try:
        for up in (False, True):
            C, H, W = (Cs, geom.Hs, geom.Ws) if up else (Cb, Hb, Wb)        # the consumer's input
            Co = Cb if up else Cs
            x = dev(synth_vec((B, C, H, W), 0.371, 0.2) * 1.3 + 0.1)
            w = dev(synth_vec((Cs, Cb, k, k), 0.6180, 0.7) * (1.0 / np.sqrt(C * k * k)))
            bias = dev(synth_vec((Co,), 1.1, 0.3) * 0.1)
            gamma, beta = dev(1.0 + 0.3 * synth_vec((C,), 2.1, 0.1)), dev(0.3 * synth_vec((C,), 2.9, 0.6))
            stats = torch.zeros(2 * C, device='cuda', dtype=torch.float64)
            ops.bn_stats(x, stats)
            fn = ops.conv_up if up else ops.conv_down
            res = []
            for fused in (False, True):
                rm, rv = dev(synth_vec((C,), 0.5, 0.5)), dev(synth_vec((C,), 0.7, 0.1).abs() + 0.5)
                nbt = torch.tensor(3, device='cuda', dtype=torch.int64)
                vec = [torch.full((C,), float('nan'), device='cuda') for _ in range(4)]
                src = ops.bn_src(stats, B * H * W, gamma, beta, 1e-5, 0.1, rm, rv, nbt, *vec)
                if fused:
                    y = fn(geom, x, w, bias, ops.PGV_ACT_LEAKY_RELU, 0.1, in_bn=src)
                else:
                    ops.bn_src_finalize(src)
                    y = fn(geom, x, w, bias, ops.PGV_ACT_LEAKY_RELU, 0.1, in_scale=vec[0], in_shift=vec[1])
                res.append([y, rm, rv, nbt] + vec)
            for a, b in zip(*res):
                assert torch.equal(a, b)
            assert res[1][3].item() == 4
            # the statistics OUTPUT as partial copies (PGV_STATS_COPIES): the copies add up to the one-copy result, and
            # a finalize over the copies gives the same vectors as over their sum
            Cout = Cb if up else Cs
            st1 = torch.zeros(2 * Cout, device='cuda', dtype=torch.float64)
            st8 = torch.zeros(ops.CLS_COPIES * 2 * Cout, device='cuda', dtype=torch.float64)
            y1 = fn(geom, x, w, bias, ops.PGV_ACT_LEAKY_RELU, 0.1, stats=st1, prezeroed=True)
            y8 = fn(geom, x, w, bias, ops.PGV_ACT_LEAKY_RELU, 0.1, stats=st8, prezeroed=True, stats_copies=True)
            assert torch.equal(y1, y8)
            s8 = st8.view(ops.CLS_COPIES, 2 * Cout).sum(0)
            assert (s8 - st1).abs().max().item() <= 1e-9 * st1.abs().max().item()
            n_o = y1.numel() // Cout
            g_o, b_o = dev(1.0 + 0.3 * synth_vec((Cout,), 2.1, 0.1)), dev(0.3 * synth_vec((Cout,), 2.9, 0.6))
            va = [torch.empty(Cout, device='cuda') for _ in range(4)]
            vb = [torch.empty(Cout, device='cuda') for _ in range(4)]
            ops.bn_src_finalize(ops.bn_src(s8.contiguous(), n_o, g_o, b_o, 1e-5, 0.1, None, None, None, *va))
            ops.bn_src_finalize(ops.bn_src(st8, n_o, g_o, b_o, 1e-5, 0.1, None, None, None, *vb, stats_copies=ops.CLS_COPIES))
            for a, b in zip(va, vb):
                assert (a - b).abs().max().item() <= 1e-6 * max(1.0, a.abs().max().item())
        # the Dropout in front of the encoder's Linear: same draw, same values
        C, HW = Cb, 24
        x = dev(synth_vec((B, C, 4, 6), 0.371, 0.2))
        stats = torch.zeros(2 * C, device='cuda', dtype=torch.float64)
        ops.bn_stats(x, stats)
        gamma, beta = dev(1.0 + 0.3 * synth_vec((C,), 2.1, 0.1)), dev(0.3 * synth_vec((C,), 2.9, 0.6))
        res = []
        for fused in (False, True):
            rng = DeviceRNG(torch.device('cuda'), seed=5)
            vec = [torch.full((C,), float('nan'), device='cuda') for _ in range(4)]
            rm, rv = torch.zeros(C, device='cuda'), torch.ones(C, device='cuda')
            src = ops.bn_src(stats, B * HW, gamma, beta, 1e-5, 0.1, rm, rv, None, *vec)
            if fused:
                y, _ = rng.dropout_nomask(0.3, x, 2, in_bn=src)
            else:
                ops.bn_src_finalize(src)
                y, _ = rng.dropout_nomask(0.3, x, 2, vec[0], vec[1])
            res.append([y, rm, rv] + vec)
        for a, b in zip(*res):
            assert torch.equal(a, b)
    finally:
        lib.pgv_set_kernel_policy(0)


@pytest.mark.parametrize("policy", [0, 3])
@pytest.mark.parametrize("case", [(8, 16, 4, 2, 2, 129, 174, 6), (16, 32, 4, 2, 2, 65, 88, 9), (32, 64, 4, 2, 2, 33, 45, 12),
                                  (64, 128, 4, 2, 2, 17, 23, 5), (1, 8, 5, 2, 2, 257, 347, 3), (3, 5, 4, 2, 2, 10, 13, 4),
                                  (256, 512, 4, 2, 2, 5, 7, 6), (512, 2048, 1, 1, 0, 3, 4, 8)])
def test_bn_backward_without_a_pass(ops, case, policy):
    """The pass-free backward of [conv -> LeakyReLU -> BatchNorm] under a consumer block (model/layer.py:21-26): tap
    sums + weight gradient of the consumer -> pgv_bn_bwd_coef -> consumer's input gradient with pgv_bwd_fuse, against
    float64 autograd of the reference arithmetic (LeakyReLU, train-mode batch_norm, conv2d / conv_transpose2d): the
    gradient of the pre-activation tensor, of the bias (its sum), of gamma and beta."""
    from preset_gen_vae_amd import _lib
    Cb, Cs, k, s, p, Hb, Wb, B = case
    geom = ops.ConvGeom(Cb, Cs, k, s, p, Hb, Wb)
    Hs, Ws = geom.Hs, geom.Ws
    oph, opw = Hb - ((Hs - 1) * s - 2 * p + k), Wb - ((Ws - 1) * s - 2 * p + k)
    w = synth_vec((Cs, Cb, k, k), 0.6180, 0.7) * (1.0 / np.sqrt(Cb * k * k / (s * s)))
    lib = _lib.load()
    lib.pgv_set_kernel_policy(policy)
    try:
        for lower_is_big in (True, False):
            C, H, W = (Cb, Hb, Wb) if lower_is_big else (Cs, Hs, Ws)
            Co, Ho, Wo = (Cs, Hs, Ws) if lower_is_big else (Cb, Hb, Wb)
            y = (synth_vec((B, C, H, W), 0.9137, 0.3) * 1.5 + 0.2).requires_grad_(True)   # pre-activation of block l
            gamma = (1.0 + 0.3 * synth_vec((C,), 2.1, 0.1)).requires_grad_(True)
            beta = (0.3 * synth_vec((C,), 2.9, 0.6)).requires_grad_(True)
            gy = synth_vec((B, Co, Ho, Wo), 0.7719, 1.1) + 0.03                        # gradient of block l+1's output
            a = F.leaky_relu(y, 0.1)
            o = F.batch_norm(a, None, None, gamma, beta, training=True, eps=1e-5)
            if lower_is_big:
                z = F.conv2d(o, w, None, stride=s, padding=p)
            else:
                z = F.conv_transpose2d(o, w, None, stride=s, padding=p, output_padding=(oph, opw))
            z.backward(gy)
            # device side: what the forward saved
            ad = a.detach()
            mean = ad.mean(dim=(0, 2, 3))
            rstd = 1.0 / torch.sqrt(ad.var(dim=(0, 2, 3), unbiased=False) + 1e-5)
            scale, shift = gamma.detach() * rstd, beta.detach() - mean * gamma.detach() * rstd
            a_d, gy_d, w_d = dev(ad), dev(gy), dev(w)
            sc_d, sh_d, mu_d, rs_d = dev(scale), dev(shift), dev(mean), dev(rstd)
            gw = torch.empty((Cs, Cb, k, k), device='cuda')
            if lower_is_big:
                ops.conv_wgrad(geom, a_d, gy_d, gw, big_scale=sc_d, big_shift=sh_d)
            else:
                ops.conv_wgrad(geom, gy_d, a_d, gw, small_scale=sc_d, small_shift=sh_d)
            T = ops.conv_tap_sums(geom, gy_d, not lower_is_big)
            coef = torch.empty(3 * C, device='cuda')
            gg, gbt = torch.empty(C, device='cuda'), torch.empty(C, device='cuda')
            ops.bn_bwd_coef(geom, B, lower_is_big, w_d, gw, T, sc_d, sh_d, mu_d, rs_d, B * H * W, coef, gg, gbt)
            # the same in one launch from the class sums of gy (border form of the tap sums + coefficient tail)
            cls = ops.conv_class_sums(geom, gy_d, not lower_is_big)
            T2 = torch.zeros(T.numel() + 1, device='cuda', dtype=torch.float64)
            coef2, gg2, gbt2 = torch.empty_like(coef), torch.empty_like(gg), torch.empty_like(gbt)
            ops.bn_bwd_coef_from_gy(geom, lower_is_big, gy_d, cls, T2, w_d, gw, sc_d, sh_d, mu_d, rs_d, B * H * W, coef2,
                                    gg2, gbt2, prezeroed=True)
            cscale = coef.abs().max().item()
            assert (coef2 - coef).abs().max().item() <= 2e-5 * cscale, (lower_is_big, (coef2 - coef).abs().max().item())
            assert rel_l2(gg2, gg) < 1e-4 and rel_l2(gbt2, gbt) < 1e-4
            # and as part of the consumer's weight-gradient call (pgv_conv_wgrad_coef), into an uninitialised gradient and
            # into a zeroed one (PGV_PREZEROED)
            for prior in (None, 0.0):
                gw3 = torch.full_like(gw, float('nan')) if prior is None else torch.zeros_like(gw)
                T3 = torch.zeros(ops.coef_scratch(geom, lower_is_big), device='cuda', dtype=torch.float64)
                coef3, gg3, gbt3 = torch.empty_like(coef), torch.empty_like(gg), torch.empty_like(gbt)
                req = dict(lower_is_big=lower_is_big, cls=cls, w=w_d, scale=sc_d, shift=sh_d, mean=mu_d, rstd=rs_d,
                           n=B * H * W, coef=coef3, ggamma=gg3, gbeta=gbt3, scratch=T3)
                if lower_is_big and prior is not None:
                    # the class sums of a Conv2d consumer = its bias gradient, here kept as partial copies (cls_copies)
                    spread = torch.zeros(ops.CLS_COPIES, cls.numel(), device='cuda')
                    spread[1], spread[6] = 0.25 * cls, 0.75 * cls
                    req.update(cls=spread.reshape(-1), cls_copies=ops.CLS_COPIES)
                if lower_is_big:
                    ops.conv_wgrad(geom, a_d, gy_d, gw3, big_scale=sc_d, big_shift=sh_d, prezeroed=prior is not None,
                                   coef_req=req)
                else:
                    ops.conv_wgrad(geom, gy_d, a_d, gw3, small_scale=sc_d, small_shift=sh_d, prezeroed=prior is not None,
                                   coef_req=req)
                assert (gw3 - gw).abs().max().item() <= 1e-5 * gw.abs().max().item()
                assert (coef3 - coef).abs().max().item() <= 2e-5 * cscale, (lower_is_big, prior)
                assert rel_l2(gg3, gg) < 1e-4 and rel_l2(gbt3, gbt) < 1e-4
            gb = torch.zeros(C, device='cuda')
            fuse = (a_d, coef, gb, ops.PGV_ACT_LEAKY_RELU, 0.1)
            if lower_is_big:
                g_y = ops.conv_up(geom, gy_d, w_d, None, ops.PGV_ACT_NONE, 0.0, bwd_fuse=fuse)
            else:
                g_y = ops.conv_down(geom, gy_d, w_d, None, ops.PGV_ACT_NONE, 0.0, bwd_fuse=fuse)
            # the projections are sums of n = B*H*W products g*o that cancel: error bounds relative to the sum of |g*o|
            assert rel_l2(g_y, y.grad) < 2e-5, (lower_is_big, rel_l2(g_y, y.grad))
            n = B * H * W
            gscale = y.grad.abs().mean().item() * n          # ~ sum |g| per channel
            assert (gbt.cpu().double() - beta.grad).abs().max().item() <= 2e-6 * gscale * 4
            assert (gg.cpu().double() - gamma.grad).abs().max().item() <= 2e-6 * gscale * 16
            assert (gb.cpu().double() - y.grad.sum(dim=(0, 2, 3))).abs().max().item() <= 2e-6 * gscale * 4
    finally:
        lib.pgv_set_kernel_policy(0)


@pytest.mark.parametrize("shape", [(3, 1, 257, 347), (5, 1, 129, 128), (2, 1, 131, 174)])
def test_sqerr_act_bwd_with_class_sums(ops, shape):
    """pgv_sqerr_act_bwd_cls = pgv_sqerr_act_bwd plus the sums of g_y by (row parity, column parity) class."""
    B, C, H, W = shape
    a = dev(torch.clamp(synth_vec(shape, 0.713, 0.2) * 1.4, -1.0, 1.0))
    x = dev(synth_vec(shape, 0.377, 0.9))
    gl = torch.tensor(0.7, device='cuda')
    scale = 1.0 / a.numel()
    g1, gb1, l1 = torch.empty_like(a), torch.zeros(1, device='cuda'), torch.zeros((), device='cuda')
    ops.sqerr_act_bwd(a, x, gl, scale, ops.PGV_ACT_HARDTANH, 0.0, g1, gb1, prezeroed=True, loss_acc=l1)
    g2, gb2, l2 = torch.empty_like(a), torch.zeros(1, device='cuda'), torch.zeros((), device='cuda')
    cls = torch.zeros(ops.CLS_COPIES * 4, device='cuda')
    ops.sqerr_act_bwd(a, x, gl, scale, ops.PGV_ACT_HARDTANH, 0.0, g2, gb2, prezeroed=True, loss_acc=l2, cls=cls)
    cls = cls.view(ops.CLS_COPIES, 4).sum(0)   # (partial copies per XCD)
    assert torch.equal(g1, g2)
    ref = torch.stack([g1.double()[:, :, r::2, c::2].sum() for r in range(2) for c in range(2)])
    tol = 2e-6 * g1.double().abs().sum().item() + 1e-12
    assert (cls.double() - ref).abs().max().item() <= tol
    assert abs(gb2.item() - g1.double().sum().item()) <= tol and abs(l1.item() - l2.item()) <= 1e-6 * abs(l1.item())


@pytest.mark.parametrize("case", [(8, 16, 4, 2, 2, 129, 174, 5), (16, 32, 4, 2, 2, 65, 88, 7), (32, 64, 4, 2, 2, 33, 45, 9),
                                  (1, 8, 5, 2, 2, 257, 347, 3), (3, 5, 4, 2, 2, 10, 13, 2)])
def test_conv_wgrad_without_workspace(ops, case):
    """pgv_conv_wgrad with a NULL (or too small) workspace: the wave-specialised kernels need one partial gradient per
    workgroup there, so the call must fall back to the kernels that flush with float atomics (pgv_hip.h) - same result,
    overwrite and PGV_PREZEROED accumulate semantics."""
    import ctypes
    from preset_gen_vae_amd import _lib
    Cb, Cs, k, s, p, Hb, Wb, B = case
    big, small, w, bias_s, bias_b, sc_b, sh_b, sc_s, sh_s, Hs, Ws = _conv_inputs(case)
    geom = ops.ConvGeom(Cb, Cs, k, s, p, Hb, Wb)
    wv = w.clone().requires_grad_(True)
    F.conv2d(_affine(big, sc_b, sh_b), wv, None, stride=s, padding=p).backward(small)
    lib = _lib.load()
    bd, sd, scd, shd = dev(big), dev(small), dev(sc_b), dev(sh_b)
    st = torch.cuda.current_stream().cuda_stream
    for ws_bytes in (0, 64):
        ws = torch.empty(16, device='cuda') if ws_bytes else None
        gw = torch.full((Cs, Cb, k, k), 7.0, device='cuda')
        _lib.check(lib.pgv_conv_wgrad(ctypes.byref(geom.desc(B)), bd.data_ptr(), scd.data_ptr(), shd.data_ptr(),
                                      sd.data_ptr(), None, None, gw.data_ptr(), None if ws is None else ws.data_ptr(),
                                      ws_bytes, st), "pgv_conv_wgrad")
        assert rel_l2(gw, wv.grad) < 5e-5          # overwritten
        _lib.check(lib.pgv_conv_wgrad(ctypes.byref(geom.desc(B, ops.PGV_PREZEROED)), bd.data_ptr(), scd.data_ptr(),
                                      shd.data_ptr(), sd.data_ptr(), None, None, gw.data_ptr(),
                                      None if ws is None else ws.data_ptr(), ws_bytes, st), "pgv_conv_wgrad")
        assert rel_l2(gw, 2 * wv.grad) < 5e-5      # accumulated into


@pytest.mark.parametrize("shape", [(256, 128), (7, 33), (2, 1024), (19, 64)])
def test_batchnorm1d_one_launch_per_direction(ops, shape):
    """pgv_bn1d_fwd / pgv_bn1d_bwd (nn.BatchNorm1d, encoder.py:86-87, train mode) against float64 torch: output, saved
    statistics, running-statistics update, all gradients."""
    B, C = shape
    x = (synth_vec(shape, 0.713, 0.2) * 1.7 + 0.3).requires_grad_(True)
    gamma = (1.0 + 0.3 * synth_vec((C,), 2.1, 0.1)).requires_grad_(True)
    beta = (0.2 * synth_vec((C,), 2.9, 0.6)).requires_grad_(True)
    rm, rv = 0.1 * synth_vec((C,), 1.3, 0.4), 1.0 + 0.2 * synth_vec((C,), 1.9, 0.8)
    g = synth_vec(shape, 0.377, 0.9)
    rm_ref, rv_ref = rm.clone(), rv.clone()
    y = F.batch_norm(x, rm_ref, rv_ref, gamma, beta, training=True, momentum=0.1, eps=1e-5)
    y.backward(g)
    xd, gd = dev(x.detach()), dev(g)
    rmd, rvd = dev(rm), dev(rv)
    nbt = torch.zeros((), device='cuda', dtype=torch.int64)
    yd = torch.empty_like(xd)
    sc, mu, rs = (torch.empty(C, device='cuda') for _ in range(3))
    ops.bn1d_fwd(xd, dev(gamma.detach()), dev(beta.detach()), 1e-5, 0.1, rmd, rvd, nbt, yd, sc, mu, rs)
    assert rel_l2(yd, y) < 1e-5 and int(nbt.item()) == 1
    assert rel_l2(rmd, rm_ref) < 1e-6 and rel_l2(rvd, rv_ref) < 1e-6
    gx, gg, gb = torch.empty_like(xd), torch.empty(C, device='cuda'), torch.empty(C, device='cuda')
    ops.bn1d_bwd(gd, xd, sc, mu, rs, gx, gg, gb)
    assert rel_l2(gx, x.grad) < 2e-5
    assert rel_l2(gg, gamma.grad) < 1e-5 and rel_l2(gb, beta.grad) < 1e-5


@pytest.mark.parametrize("shape", [(256, 64), (7, 5), (19, 20), (2, 64)])
def test_encoder_head_in_one_launch(ops, shape):
    """pgv_bn1d_reparam_fwd / _bwd = pgv_bn1d_fwd -> pgv_reparam_kl_fwd_rng and pgv_reparam_kl_bwd -> pgv_bn1d_bwd ->
    column sums, each direction as one launch: the same values (outputs identical, sums to rounding)."""
    from preset_gen_vae_amd.rng import DeviceRNG
    B, D = shape
    C = 2 * D
    x = dev(synth_vec((B, C), 0.713, 0.2) * 1.7 + 0.3)
    gamma, beta = dev(1.0 + 0.3 * synth_vec((C,), 2.1, 0.1)), dev(0.2 * synth_vec((C,), 2.9, 0.6))
    g_z, g_y = dev(synth_vec((B, D), 0.377, 0.9)), dev(0.1 * synth_vec((B, C), 0.177, 0.3))
    g_kl = torch.tensor(0.7, device='cuda')
    # separate launches
    rng = DeviceRNG(torch.device('cuda'), seed=3)
    rm, rv = dev(0.1 * synth_vec((C,), 1.3, 0.4)), dev(1.0 + 0.2 * synth_vec((C,), 1.9, 0.8))
    nbt = torch.zeros((), device='cuda', dtype=torch.int64)
    y = torch.empty_like(x)
    sc, mu, rs = (torch.empty(C, device='cuda') for _ in range(3))
    ops.bn1d_fwd(x, gamma, beta, 1e-5, 0.1, rm, rv, nbt, y, sc, mu, rs)
    z, kl, eps = rng.reparam_kl(y.view(B, 2, D), 0.25)
    g_ml = ops.reparam_kl_bwd(y.view(B, 2, D), eps, g_z, g_kl, 0.25).view(B, C) + g_y
    gx, gg, gb = torch.empty_like(x), torch.empty(C, device='cuda'), torch.empty(C, device='cuda')
    ops.bn1d_bwd(g_ml, x, sc, mu, rs, gx, gg, gb)
    # one launch each
    rng2 = DeviceRNG(torch.device('cuda'), seed=3)
    rm2, rv2 = dev(0.1 * synth_vec((C,), 1.3, 0.4)), dev(1.0 + 0.2 * synth_vec((C,), 1.9, 0.8))
    nbt2 = torch.zeros((), device='cuda', dtype=torch.int64)
    y2, sc2, mu2, rs2, z2, kl2, eps2 = rng2.bn1d_reparam(x, gamma, beta, 1e-5, 0.1, rm2, rv2, nbt2, 0.25)
    assert torch.equal(eps2, eps) and torch.equal(nbt2, nbt)
    for a, b in ((y2, y), (sc2, sc), (mu2, mu), (rs2, rs), (z2, z), (rm2, rm), (rv2, rv)):   # (sums in another order)
        assert (a - b).abs().max().item() <= 2e-6 * max(b.abs().max().item(), 1.0)
    assert torch.equal(rng2.state, rng.state)
    assert abs(kl2.item() - kl.item()) <= 1e-5 * abs(kl.item())
    gx2, gg2, gb2 = torch.empty_like(x), torch.empty(C, device='cuda'), torch.empty(C, device='cuda')
    cs = torch.full((C,), float('nan'), device='cuda')
    ops.bn1d_reparam_bwd(g_z, g_kl, g_y, y2, eps2, x, sc2, mu2, rs2, 0.25, gx2, gg2, gb2, colsum=cs)
    scale_g = g_ml.abs().max().item()
    assert (gx2 - gx).abs().max().item() <= 1e-5 * scale_g * sc.abs().max().item()
    assert rel_l2(gg2, gg) < 1e-5 and rel_l2(gb2, gb) < 1e-5
    assert (cs.double() - gx2.double().sum(0)).abs().max().item() <= 1e-5 * gx2.abs().max().item() * B
    # without g_y / g_z (only the Dkl gradient), into an accumulating column sum
    cs3 = torch.ones(C, device='cuda')
    gx3 = torch.empty_like(x)
    ops.bn1d_reparam_bwd(None, g_kl, None, y2, eps2, x, sc2, mu2, rs2, 0.25, gx3, None, None, colsum=cs3, colsum_accumulate=True)
    g_ml3 = ops.reparam_kl_bwd(y.view(B, 2, D), eps, None, g_kl, 0.25).view(B, C)
    gx_ref3 = torch.empty_like(x)
    ops.bn1d_bwd(g_ml3, x, sc, mu, rs, gx_ref3, gg, gb)
    assert (gx3 - gx_ref3).abs().max().item() <= 1e-5 * g_ml3.abs().max().item() * sc.abs().max().item()
    assert (cs3.double() - 1.0 - gx3.double().sum(0)).abs().max().item() <= 1e-5 * max(gx3.abs().max().item() * B, 1.0)


def test_conv_desc_validation(ops):
    from preset_gen_vae_amd import _lib
    geom = ops.ConvGeom(2, 3, 4, 2, 2, 9, 9)
    geom.Hs += 1   # inconsistent geometry must be refused with an error code, not executed
    x = torch.zeros((1, 2, 9, 9), device='cuda')
    w = torch.zeros((3, 2, 4, 4), device='cuda')
    with pytest.raises(RuntimeError, match="inconsistent"):
        ops.conv_down(geom, x, w, None, 0, 0.0)
    with pytest.raises(RuntimeError, match="ROCm device"):
        ops.conv_down(ops.ConvGeom(2, 3, 4, 2, 2, 9, 9), x.cpu(), w, None, 0, 0.0)
    assert _lib.load().pgv_abi_version() == 16


def test_empty_batch(ops):
    geom = ops.ConvGeom(2, 3, 4, 2, 2, 9, 9)
    x = torch.zeros((0, 2, 9, 9), device='cuda')
    w = torch.zeros((3, 2, 4, 4), device='cuda')
    out = ops.conv_down(geom, x, w, None, 0, 0.0)
    assert out.shape == (0, 3, geom.Hs, geom.Ws)


@pytest.mark.parametrize("shape", [(4, 16, 65 * 88), (3, 7, 33 * 45), (6, 2048, 12), (256, 128, 1), (2, 8, 129 * 174),
                                   (19, 2048, 12), (9, 512, 35), (33, 128, 108), (12, 64, 391), (8, 130, 12)])
def test_batchnorm_pieces(ops, shape):
    B, C, HW = shape
    a = synth_vec((B, C, HW), 0.831, 0.2) * 1.3 + 0.4 * synth_vec((1, C, 1), 1.9, 0.3)
    g_o = synth_vec((B, C, HW), 0.557, 1.2)
    gamma, beta = 1.0 + 0.1 * synth_vec((C,), 2.3, 0.1), 0.1 * synth_vec((C,), 3.3, 0.2)
    rm, rv = 0.1 * synth_vec((C,), 4.1, 0.3), 1.0 + 0.2 * synth_vec((C,), 5.1, 0.4)
    # oracle: torch batch_norm in float64 with autograd
    a_r = a.clone().requires_grad_(True)
    gam_r, bet_r = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm_r, rv_r = rm.clone(), rv.clone()
    lre = F.leaky_relu(a_r, 0.1)
    o = F.batch_norm(lre, rm_r, rv_r, gam_r, bet_r, training=True, momentum=0.1, eps=1e-5)
    o.backward(g_o)
    act = lre.detach()
    # HIP
    d_act = dev(act)
    stats = torch.empty(2 * C, device='cuda', dtype=torch.float64)
    ops.bn_stats(d_act, stats)
    vec = torch.empty(4 * C, device='cuda')
    scale, shift, mean, rstd = vec[:C], vec[C:2 * C], vec[2 * C:3 * C], vec[3 * C:]
    d_rm, d_rv = dev(rm), dev(rv)
    nbt = torch.full((), 41, device='cuda', dtype=torch.int64)
    ops.bn_finalize(stats, B * HW, dev(gamma), dev(beta), 1e-5, 0.1, d_rm, d_rv, scale, shift, mean, rstd,
                    num_batches_tracked=nbt)
    assert int(nbt) == 42   # nn.BatchNorm's counter advances in the same launch
    out = ops.affine_nchw(d_act, scale, shift)
    assert rel_l2(out, o) < 1e-5
    assert rel_l2(d_rm, rm_r) < 1e-5 and rel_l2(d_rv, rv_r) < 1e-5
    red = torch.empty(2 * C, device='cuda', dtype=torch.float64)
    d_go = dev(g_o)
    ops.bn_bwd_reduce(d_go, d_act, mean, rstd, red)
    assert rel_l2(red[C:], gam_r.grad) < 1e-4 and rel_l2(red[:C], bet_r.grad) < 1e-4
    g_y = torch.empty_like(d_go)
    gbias = torch.empty(C, device='cuda')
    ggamma, gbeta = torch.empty(C, device='cuda'), torch.empty(C, device='cuda')
    ops.act_bn_bwd(d_go, d_act, scale, mean, rstd, red, ops.PGV_ACT_LEAKY_RELU, 0.1, g_y, gbias, ggamma=ggamma,
                   gbeta=gbeta)
    assert rel_l2(g_y, a_r.grad) < 1e-4
    assert rel_l2(gbias, a_r.grad.sum(dim=(0, 2))) < 1e-3 or a_r.grad.sum(dim=(0, 2)).abs().max() < 1e-6
    assert torch.equal(ggamma, red[C:].float()) and torch.equal(gbeta, red[:C].float())
    # PGV_PREZEROED: the call accumulates into what the buffer holds (the caller cleared it once for many calls)
    red2 = red.clone()
    ops.bn_bwd_reduce(d_go, d_act, mean, rstd, red2, prezeroed=True)
    assert rel_l2(red2, 2 * red) < 1e-12
    gb2 = gbias.clone()
    ops.act_bn_bwd(d_go, d_act, scale, mean, rstd, red, ops.PGV_ACT_LEAKY_RELU, 0.1, g_y, gb2, prezeroed=True)
    assert rel_l2(gb2, 2 * gbias) < 1e-5 or gbias.abs().max() < 1e-6
    # eval-mode affine
    ops.bn_eval_affine(dev(gamma), dev(beta), dev(rm), dev(rv), 1e-5, scale, shift)
    ref = F.batch_norm(act, rm.clone(), rv.clone(), gamma, beta, training=False, eps=1e-5)
    assert rel_l2(ops.affine_nchw(d_act, scale, shift), ref) < 1e-5


@pytest.mark.parametrize("mnk", [(256, 128, 25024), (256, 25024, 64), (16, 128, 24576), (5, 7, 3), (64, 64, 64),
                                 (2, 24576, 64), (130, 70, 33),
                                 # the fragment-streaming kernels (gemm_frag.hip): z = 512 extents, K splits that do not
                                 # divide the chunk count, single-chunk-pair K, ragged job counts
                                 (256, 1024, 25024), (256, 25024, 512), (96, 192, 4112), (32, 64, 32), (160, 320, 48)])
def test_linear_gemm(ops, mnk):
    M, N, K = mnk
    x = synth_vec((M, K), 0.771, 0.3)
    w = synth_vec((N, K), 0.613, 0.8) / np.sqrt(K)
    b = synth_vec((N,), 1.1, 0.2)
    gy = synth_vec((M, N), 0.913, 0.5)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y = F.linear(xr, wr, br)
    y.backward(gy)
    dx, dw, db, dgy = dev(x), dev(w), dev(b), dev(gy)
    assert rel_l2(ops.linear_fwd(dx, dw, db), y) < 1e-5
    assert rel_l2(ops.linear_dgrad(dgy, dw), xr.grad) < 1e-5
    gw = torch.empty_like(dw)
    ops.linear_wgrad(dgy, dx, gw)
    assert rel_l2(gw, wr.grad) < 1e-5
    gb = torch.empty(N, device='cuda')
    ops.colsum(dgy, gb)
    assert rel_l2(gb, br.grad) < 1e-5


@pytest.mark.parametrize("mnk", [(64, 128, 24576), (256, 1024, 3200), (256, 192, 4096), (512, 256, 1056),
                                 # round 6: the fragment-streaming kernels with 32-deep bf16 chunks (gemm_frag.hip) - BASELINE config
                                 # 2's extents (z = 512 on 12 288 features), the z = 64 ones, K splits that do not divide the chunk
                                 # count, two-chunk K, ragged job counts
                                 (256, 1024, 12288), (256, 12288, 512), (256, 128, 25024), (256, 25024, 64), (96, 192, 4128),
                                 (32, 64, 64), (160, 320, 96)])
def test_linear_gemm_bf16_operand_mode(ops, mnk):
    """pgv_gemm flags = PGV_COMPUTE_BF16: the three nn.Linear products with bf16-rounded operands (M = 256 / 512: the
    256-row tiles of round 4, 128- and 64-wide; gemm_frag.hip's bf16 form where it covers the shape, and gemm.hip's tiles
    for the same shape with the fragment kernels switched off)."""
    M, N, K = mnk
    x = synth_vec((M, K), 0.771, 0.3)
    w = synth_vec((N, K), 0.613, 0.8) / np.sqrt(K)
    b = synth_vec((N,), 1.1, 0.2)
    gy = synth_vec((M, N), 0.913, 0.5)
    dx, dw, db, dgy = dev(x), dev(w), dev(b), dev(gy)
    from preset_gen_vae_amd import _lib
    ops.set_compute_dtype('bf16')
    try:
        for tiles in ((0, 1, 2, 3) if M % 256 == 0 else (0, 2, 3)):   # (1: the 256-row tiles, off by default - slower, gemm.hip;
            _lib.load().pgv_dbg_set_gemm_tiles(tiles & 1 if tiles < 3 else 0)   # 2: everything on gemm.hip's LDS tiles; 3: every
            _lib.load().pgv_dbg_set_gemm_variant({2: 1024, 3: 4096}.get(tiles, 0))   # covered shape on gemm_frag.hip's bf16 form)
            y = ops.linear_fwd(dx, dw, db)
            assert rel_l2(y, _bf16(x) @ _bf16(w).t() + b.float().double()) < 1e-5
            assert rel_l2(y, F.linear(x, w, b)) > 2e-4
            assert rel_l2(ops.linear_dgrad(dgy, dw), _bf16(gy) @ _bf16(w)) < 1e-5
            gw = torch.empty_like(dw)
            ops.linear_wgrad(dgy, dx, gw)
            assert rel_l2(gw, _bf16(gy).t() @ _bf16(x)) < 1e-5
    finally:
        _lib.load().pgv_dbg_set_gemm_tiles(0)
        _lib.load().pgv_dbg_set_gemm_variant(0)
        ops.set_compute_dtype('fp32')


def test_reparam_kl_and_sqerr(ops):
    from oracle import vae_oracle as vo
    B, D = 37, 64
    ml = (synth_vec((B, 2, D), 0.77, 0.1) * 0.8).requires_grad_(True)
    eps = synth_vec((B, D), 1.31, 0.4) * 1.2
    gz = synth_vec((B, D), 0.41, 0.9)
    z = vo.reparametrize(ml, eps, True)
    kl = vo.gaussian_dkl(ml[:, 0], ml[:, 1], normalize=True)
    (z * gz).sum().backward(retain_graph=True)
    g_from_z = ml.grad.clone()
    ml.grad = None
    (kl * 0.7).backward()
    g_from_kl = ml.grad.clone()
    d_ml, d_eps = dev(ml.detach()), dev(eps)
    z_hip, _ = ops.reparam_kl_fwd(d_ml, d_eps, 0.0)
    assert rel_l2(z_hip, z) < 1e-6
    z_eval, kl_hip = ops.reparam_kl_fwd(d_ml, None, 1.0 / (B * D))
    assert rel_l2(z_eval, ml[:, 0]) == 0 or rel_l2(z_eval, ml[:, 0]) < 1e-7
    assert abs(kl_hip.item() - kl.item()) < 1e-5 * abs(kl.item())
    assert rel_l2(ops.reparam_kl_bwd(d_ml, d_eps, dev(gz), None, 0.0), g_from_z) < 1e-5
    gk = torch.tensor(0.7, device='cuda')
    assert rel_l2(ops.reparam_kl_bwd(d_ml, None, None, gk, 1.0 / (B * D)), g_from_kl) < 1e-5
    # squared error, with and without the Hardtanh gate
    n = 2 * 257 * 347
    xhat = (synth_vec((n,), 0.31, 0.2) * 1.4).clamp(-1, 1).requires_grad_(True)
    x = synth_vec((n,), 0.47, 0.7)
    loss = F.mse_loss(xhat, x)
    (loss * 1.7).backward()
    l_hip = ops.sqerr_fwd(dev(xhat.detach()), dev(x), 1.0 / n)
    assert abs(l_hip.item() - loss.item()) < 1e-5 * loss.item()
    g = ops.sqerr_bwd(dev(xhat.detach()), dev(x), torch.tensor(1.7, device='cuda'), 1.0 / n)
    assert rel_l2(g, xhat.grad) < 1e-5
    gate = ((xhat.detach() > -1) & (xhat.detach() < 1)).double()
    g = ops.sqerr_bwd(dev(xhat.detach()), dev(x), torch.tensor(1.7, device='cuda'), 1.0 / n, hardtanh=True)
    assert rel_l2(g, xhat.grad * gate) < 1e-5


def test_adam_matches_torch_semantics(ops):
    from oracle import vae_oracle as vo
    from preset_gen_vae_amd import optim
    n = 100003
    p0, g1, g2 = synth_vec((n,), 0.3, 0.1), synth_vec((n,), 0.7, 0.2) * 1e-2, synth_vec((n,), 0.9, 0.3) * 1e-2
    param = torch.nn.Parameter(dev(p0))
    flat = optim.FlatParams([param])
    opt = optim.FusedAdam(flat, lr=2e-4, betas=(0.9, 0.999), weight_decay=1e-4)
    p, m, v = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
    for t, g in enumerate((g1, g2), start=1):
        flat.flat_grad[:n].copy_(dev(g))
        opt.step()
        p, m, v = vo.adam_update(p, g, m, v, t, 2e-4, (0.9, 0.999), 1e-8, 1e-4)
    assert (param.detach().cpu().double() - p).abs().max().item() < 2e-7
    assert rel_l2(param.detach() - dev(p0), p - p0) < 1e-4
    # cross-check against torch.optim.Adam itself
    tp = torch.nn.Parameter(p0.clone())
    topt = torch.optim.Adam([tp], lr=2e-4, betas=(0.9, 0.999), weight_decay=1e-4)
    for g in (g1, g2):
        tp.grad = g.clone()
        topt.step()
    assert (tp.detach() - p).abs().max().item() < 1e-12


def test_rng_distributions(ops):
    from preset_gen_vae_amd.rng import DeviceRNG
    rng = DeviceRNG(torch.device('cuda'), seed=42)
    m1 = rng.dropout_mask(0.3, (256, 24576))
    m2 = rng.dropout_mask(0.3, (256, 24576))
    keep = (m1 > 0).float().mean().item()
    assert abs(keep - 0.7) < 2e-3
    vals = torch.unique(m1)
    assert vals.numel() == 2 and abs(vals.max().item() - 1 / 0.7) < 1e-6 and vals.min().item() == 0
    assert (m1 != m2).float().mean().item() > 0.3          # the stream advances between calls
    e = rng.normal((256, 4096))
    assert abs(e.mean().item()) < 5e-3 and abs(e.std().item() - 1.0) < 5e-3
    assert abs((e ** 4).mean().item() - 3.0) < 0.1
    rng_b = DeviceRNG(torch.device('cuda'), seed=42)
    assert torch.equal(rng_b.dropout_mask(0.3, (256, 24576)), m1)   # reproducible from the seed
    # the fused forward (mask drawn inside the multiply kernel) is the same draw from the same state
    rng_c = DeviceRNG(torch.device('cuda'), seed=42)
    for shape in ((256, 24576), (3, 1001)):
        x = torch.randn(*shape, device='cuda')
        y, m = rng_c.dropout(0.3, x)
        ref = DeviceRNG(torch.device('cuda'), seed=42)
        if shape != (256, 24576):
            ref.dropout_mask(0.3, (256, 24576))                     # advance the reference stream the same way
        mr = ref.dropout_mask(0.3, shape)
        assert torch.equal(m.view(*shape), mr) and torch.equal(y, x * mr)
    # eps drawn inside the reparameterisation kernel = the draw normal() makes from the same state
    rng_e, ref = DeviceRNG(torch.device('cuda'), seed=11), DeviceRNG(torch.device('cuda'), seed=11)
    ml = torch.randn(37, 2, 29, device='cuda') * 0.5
    z, kl, eps = rng_e.reparam_kl(ml, 0.25)
    eps_ref = ref.normal((37, 29))
    assert torch.equal(eps, eps_ref)
    z_ref, kl_ref = ops.reparam_kl_fwd(ml, eps_ref, 0.25)
    assert torch.equal(z, z_ref) and abs(kl.item() - kl_ref.item()) <= 1e-6 * abs(kl_ref.item())
    buf = torch.zeros(1, device='cuda')
    _, kl2, _ = DeviceRNG(torch.device('cuda'), seed=11).reparam_kl(ml, 0.25, kl=buf)
    assert kl2.data_ptr() == buf.data_ptr() and abs(kl2.item() - kl_ref.item()) <= 1e-6 * abs(kl_ref.item())
    # ... and so is the mask-free form the train step uses: forward draws and applies, backward regenerates the mask from
    # the saved copy of the state - also after the generator itself has moved on; optional per-channel affine on the way in
    for shape in ((256, 64, 16, 24), (16, 64, 17, 23), (3, 5, 7, 11), (4, 3, 2, 2), (2, 1001)):
        rng_d, ref = DeviceRNG(torch.device('cuda'), seed=7), DeviceRNG(torch.device('cuda'), seed=7)
        x = torch.randn(*shape, device='cuda')
        affine = len(shape) == 4
        sc = torch.rand(shape[1], device='cuda') + 0.5 if affine else None
        sh = torch.randn(shape[1], device='cuda') if affine else None
        y, saved = rng_d.dropout_nomask(0.3, x, 2, sc, sh)
        mr = ref.dropout_mask(0.3, shape, 2)
        xa = _affine_fma(x.cpu(), sc.cpu(), sh.cpu()).cuda() if affine else x
        assert torch.equal(y, xa * mr)
        rng_d.dropout_mask(0.3, (1000,))                               # the generator moves on
        g = torch.randn(*shape, device='cuda')
        assert torch.equal(ops.dropout_bwd(saved, 2, 0.3, g), g * mr)
        if affine:   # ... with the BatchNorm-backward projections of the result against a saved activation on the way
            a_s = torch.randn(*shape, device='cuda')
            mu_c, rs_c = torch.randn(shape[1], device='cuda') * 0.1, torch.rand(shape[1], device='cuda') + 0.5
            red = torch.zeros(2 * shape[1], device='cuda', dtype=torch.float64)
            gx = ops.dropout_bwd_bn_reduce(saved, 2, 0.3, g, a_s, mu_c, rs_c, red, prezeroed=True)
            ref_red = torch.zeros_like(red)
            ops.bn_bwd_reduce(g * mr, a_s, mu_c, rs_c, ref_red, prezeroed=True)
            assert torch.equal(gx, g * mr)
            assert (red - ref_red).abs().max().item() <= 1e-6 * (g * mr).abs().sum().item()
        if len(shape) == 2 or shape[0] == 256:   # ... with the column sums of the result (the Linear bias gradient)
            g2, m2 = g.reshape(shape[0], -1), mr.reshape(shape[0], -1)
            cs = torch.full((g2.shape[1],), 0.5, device='cuda')
            gx = ops.dropout_bwd(saved, 2, 0.3, g2, colsum=cs, prezeroed=True)
            ref_cs = (g2 * m2).double().sum(0) + 0.5
            assert torch.equal(gx, g2 * m2) and (cs.double() - ref_cs).abs().max().item() <= 1e-5 * ref_cs.abs().max().item()


@pytest.mark.parametrize("kind", ["conv", "tconv"])
def test_block_chain_bf16_vs_oracle(ops, kind):
    """Two product blocks chained (so the second one folds the first one's BatchNorm into its loader and rounds the
    NORMALISED operand) in PGV_COMPUTE_BF16 mode, forward and backward, against the oracle's blocks run with
    operand_precision('bf16') on the same inputs.  Short chains on purpose: operand rounding is discontinuous, so
    float32 noise upstream moves a few operands across a bf16 boundary downstream (measured: +1 decade per block) and
    an 8-block end-to-end comparison can only be statistical (tests/test_gpu_vae.py)."""
    from oracle import vae_oracle as vo
    from preset_gen_vae_amd.model import layer
    import torch.nn as nn
    B = 3
    if kind == 'conv':
        rows = [('enc2', 8, 16, 4, 2, 2, True), ('enc3', 16, 32, 4, 2, 2, True)]
        blocks = [layer.Conv2D(r[1], r[2], [4, 4], [2, 2], 2, [1, 1], activation=nn.LeakyReLU(0.1), name_prefix=r[0],
                               batch_norm='after') for r in rows]
        x = synth_vec((B, 8, 65, 88), 0.613, 0.4) * 1.3
        scope = 'enc.'
    else:
        rows = [('dec6', 32, 16, 4, 2, 2, (1, 0), True), ('dec7', 16, 8, 4, 2, 2, (1, 0), True)]
        blocks = [layer.TConv2D(r[1], r[2], [4, 4], [2, 2], 2, output_padding=list(r[6]),
                                activation=nn.LeakyReLU(0.1), name_prefix=r[0], batch_norm='after') for r in rows]
        x = synth_vec((B, 32, 33, 45), 0.613, 0.4) * 1.3
        scope = 'dec.'
    sd = {}
    for i, blk in enumerate(blocks):
        for k, v in blk.state_dict().items():
            if v.dtype != torch.long:
                v = synth_vec(tuple(v.shape), 0.37 + 0.11 * i + 0.01 * len(k), 0.3)
                v = (v * (1.0 / np.sqrt(v[0].numel())) if v.dim() == 4 else
                     (1.0 + 0.2 * v if k.endswith('bn.weight') else (0.5 + 0.3 * v.abs() if 'var' in k else 0.1 * v)))
            sd[scope + k] = v.float()
        blk.load_state_dict({k: sd[scope + k] for k in blk.state_dict()})
        blocks[i] = blk.float().cuda().train()
    block_fn = vo.conv_block if kind == 'conv' else vo.tconv_block
    gy = None

    def oracle(dtype):
        nonlocal gy
        xo = x.float().to(dtype).clone().requires_grad_(True)
        po = {k: v.to(dtype).clone().requires_grad_(True) for k, v in sd.items()
              if v.dtype != torch.long and 'running' not in k}
        full = {k: (v if v.dtype == torch.long else v.to(dtype)) for k, v in sd.items()}
        full.update(po)
        with vo.operand_precision('bf16'):
            h = xo
            for r in rows:
                h = block_fn(h, full, r, scope, True, {})
            if gy is None:
                gy = synth_vec(tuple(h.shape), 0.877, 0.2).float()
            h.backward(gy.to(dtype))
        return h.detach(), xo.grad, {k: v.grad for k, v in po.items()}

    # the oracle on the same float32 inputs, evaluated in float32 and in float64: their distance is the arithmetic's
    # own sensitivity to float32-level noise (operands crossing a bf16 boundary), the unit of the tolerances below
    h, gx, gp = oracle(torch.float32)
    h64, gx64, gp64 = oracle(torch.float64)
    xd = dev(x).requires_grad_(True)
    ops.set_compute_dtype('bf16')
    try:
        y = layer.run_stack(xd, [b._pgv_block for b in blocks], True)
        y.backward(dev(gy))
        torch.cuda.synchronize()
    finally:
        ops.set_compute_dtype('fp32')
    assert rel_l2(y, h) < 3 * max(rel_l2(h64, h), 1e-5)
    assert rel_l2(xd.grad, gx) < 3 * max(rel_l2(gx64, gx), 2e-5)
    for blk in blocks:
        for k, v in blk.named_parameters():
            ref = gp[scope + k]
            if ref.abs().max() < 1e-7:      # (a zero gradient has no relative error)
                continue
            # (factor 6 for the parameter gradients since round 4.  The unit is the distance between TWO realisations of
            # the same chaotic arithmetic - the oracle in float32 and in float64 - and the kernels are a third one: their
            # sums run in other orders (tiles, K = 32 MFMA steps, atomics), which moves float32 partial sums by an ulp and
            # other downstream operands across a bf16 boundary than the oracle's own rounding does.  Measured 3.2 - 4.4 x
            # on enc2conv.bias / enc2bn.weight - sums over 40 k such elements - at an absolute level of 1e-3, i.e. half a
            # bf16 ulp; every product is checked exactly, on identical operands, in test_conv_bf16_operand_mode.)
            assert rel_l2(v.grad, ref) < 6 * max(rel_l2(gp64[scope + k], ref), 2e-5), k


def test_layer_blocks_against_reference_goldens(ops):
    """The reference's own Conv2D/TConv2D outputs (tests/golden/layers_small.npz) through the product modules."""
    from preset_gen_vae_amd.model import layer
    import torch.nn as nn
    g = load_golden('layers_small.npz')
    names = sorted({k.split('/')[0] for k in g.files})
    for name in names:
        kind = str(g[name + '/kind'])
        ci, co, k, s, p, oph, opw, bn = (int(v) for v in g[name + '/cfg'])
        if kind == 'conv':
            blk = layer.Conv2D(ci, co, [k, k], [s, s], p, [1, 1], activation=nn.LeakyReLU(0.1), name_prefix=name,
                               batch_norm=('after' if bn else None))
        elif kind == 'tconv':
            blk = layer.TConv2D(ci, co, [k, k], [s, s], p, output_padding=[oph, opw], activation=nn.LeakyReLU(0.1),
                                name_prefix=name, batch_norm=('after' if bn else None))
        else:
            conv = nn.ConvTranspose2d(ci, co, [k, k], [s, s], p)
            act = nn.Hardtanh()
            blk = nn.Sequential(conv, act)
            blk._pgv = layer._Block(conv, act, None)
        sd = {kk[len(name + '/sd_in/'):]: torch.tensor(g[kk]) for kk in g.files if kk.startswith(name + '/sd_in/')}
        blk.load_state_dict(sd)
        blk = blk.float().cuda().train()
        x = dev(torch.tensor(g[name + '/x'])).requires_grad_(True)
        y = blk(x) if kind != 'tconv_last' else layer.run_stack(x, [blk._pgv], True)
        assert rel_l2(y, torch.tensor(g[name + '/y'])) < 1e-5, name
        y.backward(dev(torch.tensor(g[name + '/gy'])))
        assert rel_l2(x.grad, torch.tensor(g[name + '/gx'])) < 1e-4, name
        for kk, v in blk.named_parameters():
            ref = torch.tensor(g[name + '/grad/' + kk])
            if ref.abs().max() < 1e-9:   # conv bias under BN: mathematically zero gradient
                assert v.grad.abs().max().item() < 1e-5, (name, kk)
            else:
                assert rel_l2(v.grad, ref) < 2e-4, (name, kk)
        for kk, v in blk.state_dict().items():
            ref = torch.tensor(g[name + '/sd_out/' + kk])
            if 'running' in kk:
                assert rel_l2(v, ref) < 1e-5, (name, kk)
            elif 'num_batches' in kk:
                assert int(v) == int(ref)
