#!/usr/bin/env python
"""Headline benchmark: spectrograms/s for one full VAE train step (zero_grad -> fwd -> MSE + beta*KL -> bwd -> Adam)
on [256,1,257,347] fp32 log-mel inputs per GPU (BASELINE.json ``metric``; workload = ``configs[1]``: the 4-layer
conv-VAE, z=64, fp32, batch 256; ``--arch speccnn8l1_bn`` runs the reference-exact 8-layer stack instead).

    python bench.py --gpus N --steps K --warmup W
    (N>1: either under python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ..., or
     bare - bench.py then starts the N ranks itself as child processes of that launcher and exits with their status)

One rank per GPU; each rank owns a 256-row shard (weak scaling), gradients are summed by RCCL all-reduce buckets.
Rank 0 prints ONE JSON line with the contract fields plus ``roofline`` (dominant kernel, measured live with HIP events
on the launch stream) and ``cpu_baseline`` (the oracle's train step on the host cores, bounded sample, N=1 only).
Inputs are synthetic: FM "Dexed-like" audio pushed once through the on-GPU STFT->mel front-end; weights random-init.
"""
import argparse
import copy
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
F32_MATRIX_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 matrix (= vector) peak; the path computes in fp32
BF16_MATRIX_PEAK_TFLOPS = 2500.0  # same guide: dense bf16 MFMA peak (--dtype bf16 only)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="rows per GPU")
    ap.add_argument("--arch", default="speccnn4l1_bn", choices=["speccnn4l1_bn", "speccnn8l1_bn"])
    ap.add_argument("--dim-z", type=int, default=64)
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16"],
                    help="operand precision of the conv / linear products (bf16 = BASELINE config 2's arithmetic: bf16 "
                         "matrix cores, fp32 accumulation and storage); the default line is fp32")
    ap.add_argument("--input", default="spectrograms", choices=["spectrograms", "audio"],
                    help="audio = BASELINE config 5: every step starts from a raw-audio minibatch [B, 88576] in HBM, the "
                         "fused STFT -> mel -> dB -> min-max kernel writes the step's input buffer (timed with the step)")
    ap.add_argument("--fp32-products", default=None, choices=["native", "bf16x6"],
                    help="fp32 mode only; default = the library's own default (ops.DEFAULT_FP32_PRODUCTS = bf16x6, what "
                         "config.train.fp32_products = None and VAETrainStep() run): bf16x6 = layers with a split-product kernel evaluate every fp32 "
                         "product as six bf16 matrix instructions on exact three-way operand splits, fp32 accumulation "
                         "(DESIGN.md 2.3; the strict fp32 parity tests run in this mode too); native = the fp32 matrix "
                         "instruction everywhere (reported under 'extra')")
    ap.add_argument("--gemm-variant", type=int, default=0,
                    help="A/B timing aid: pgv_dbg_set_gemm_variant (csrc/gemm_frag.hip; 1024 = every nn.Linear product on the "
                         "LDS-tiled kernels of gemm.hip, 4096 = the covered bf16 shapes on the fragment-streaming ones too)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of one hipGraph per step")
    ap.add_argument("--dist-mode", default="bucket-graphs", choices=["bucket-graphs", "eager", "two-graph"],
                    help="N > 1 launch mode: bucket-graphs = the captured step cut at the gradient-bucket boundaries "
                         "(k + 2 hipGraphs), each bucket's all-reduce launched between two replays on the communication "
                         "stream: replay AND overlap; eager = every kernel launched from Python, all-reduce from "
                         "gradient-ready hooks; two-graph = [fwd+bwd] and [Adam] as two hipGraphs around a non-overlapped "
                         "exchange")
    ap.add_argument("--dist-graph", action="store_true", help="alias of --dist-mode two-graph")
    ap.add_argument("--buckets", type=int, default=4, help="gradient all-reduce buckets (N > 1)")
    ap.add_argument("--latent-reg", default="bn", choices=["bn", "none"],
                    help="train.latent_flow_input_regularization (reference default 'bn', config.py:92)")
    ap.add_argument("--force-dist", action="store_true",
                    help="N = 1 only: run the N > 1 code path (gradient all-reduce over a 1-rank RCCL communicator, "
                         "launch mode of --dist-mode) - readiness evidence for the multi-GPU path on a 1-GPU box")
    ap.add_argument("--no-extra", action="store_true",
                    help="N = 1: skip the additional BASELINE.json configurations reported under 'extra'")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=16)
    args = ap.parse_args()
    if args.dist_graph:
        args.dist_mode = 'two-graph'
    return args


def synth_spectrograms(B, device, seed):
    """[B,1,257,347] min-max normalised log-mel spectrograms of seeded FM voices, via the HIP
    front-end; 16 distinct voices per rank are tiled to B rows with a per-row gain so rows differ."""
    from preset_gen_vae_amd.utils.audio import MelSpectrogram
    from preset_gen_vae_amd.utils.synthetic import fm_voice
    n_voices = min(B, 16)
    waves = np.stack([fm_voice(idx=seed * 16 + i) for i in range(n_voices)])
    reps = (B + n_voices - 1) // n_voices
    gains = (0.4 + 0.6 * np.random.default_rng(seed).random((reps, n_voices, 1))).astype(np.float32)
    waves = (waves[None] * gains).reshape(-1, waves.shape[1])[:B]
    mel = MelSpectrogram(1024, 256, -120.0, 257, 22050, device=device)
    mel.set_minmax_normalization(-120.0, 0.0)
    return mel.batch(torch.tensor(waves, device=device)).clamp_(-1.0, 1.0).contiguous()


def layer_ops(ae):
    """(layer name, ConvGeom args, has BatchNorm, is transposed) of every conv block, walked from the product modules."""
    out = []
    H, W = 257, 347
    for blk in ae.encoder._all_blocks():
        g = blk.geom(H, W)
        out.append((f"enc{len(out) + 1}", (g.Cb, g.Cs, g.k, g.stride, g.pad, g.Hb, g.Wb), blk.bn is not None, False))
        H, W = g.Hs, g.Ws
    dec_blocks = ae.decoder._all_blocks()
    first = 9 - len(dec_blocks)
    for i, blk in enumerate(dec_blocks):
        g = blk.geom(H, W)
        out.append((f"dec{first + i}", (g.Cb, g.Cs, g.k, g.stride, g.pad, g.Hb, g.Wb), blk.bn is not None, True))
        H, W = g.Hb, g.Wb
    return out


_FLUSH = None
_SINK = None


def _graph_of(body):
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            body()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    return graph


def _time_graph(graph, reps=3):
    graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        graph.replay()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


_FLUSH_MS = {}


def time_kernel(fn, iters=5):
    """Average device time of one fn() launch (kernel + its memset node, as the train step issues it) in ms, measured
    the way the step runs it: as nodes of a replayed hipGraph, so no host launch latency sits between nodes.  Graph A =
    iters x [stream through a 512 MB buffer, fn()], graph B = iters x [stream through the buffer]; HIP events around
    the replays on the replay stream; the result is (A - B) / iters.  The streaming pass puts every timed launch's
    operands back in HBM (not L2 / the 256 MB Infinity Cache), as inside the train step, where ~1 GB of other tensors
    pass between two uses of a tensor.  It READS the buffer (a reduction): a rewrite would leave 256 MB of dirty lines
    whose write-back competes with the timed kernel for HBM (measured: +15..30 % on the HBM-bound kernels, against
    their in-step rocprofv3 durations)."""
    global _FLUSH, _SINK
    if _FLUSH is None:
        _FLUSH = torch.ones(128 << 20, device='cuda', dtype=torch.float32)
        _SINK = torch.zeros((), device='cuda', dtype=torch.float32)
    fn()
    torch.cuda.synchronize()

    def flush():
        torch.sum(_FLUSH, dim=(0,), out=_SINK)

    def with_fn():
        for _ in range(iters):
            flush()
            fn()

    def without_fn():
        for _ in range(iters):
            flush()

    if iters not in _FLUSH_MS:
        _FLUSH_MS[iters] = _time_graph(_graph_of(without_fn))
    return max(_time_graph(_graph_of(with_fn)) - _FLUSH_MS[iters], 0.0) / iters


_PEAKS = None


def measured_peaks(device):
    """What THIS box sustains on the two roofline resources (SURVEY.md section 8d asks for measured AND nominal peaks):
    HBM reads - a 16-byte-per-lane grid-stride read reduction over 512 MB (pgv_probe_read) - and the matrix pipe - a
    dependency-free MFMA stream from one wave per SIMD on every CU, ~1 ms per launch, lane-dependent operands
    (pgv_probe_mfma: v_mfma_f32_16x16x4_f32 and v_mfma_f32_16x16x32_bf16).  Timed like the launches of the step: HIP events
    around graph replays.  Measured once per process."""
    global _PEAKS, _FLUSH, _SINK
    if _PEAKS is not None:
        return _PEAKS
    import ctypes
    from preset_gen_vae_amd import _lib
    from preset_gen_vae_amd.ops import _stream
    lib = _lib.load()
    if _FLUSH is None:
        _FLUSH = torch.ones(128 << 20, device=device, dtype=torch.float32)
        _SINK = torch.zeros((), device=device, dtype=torch.float32)
    sink = torch.zeros(4, device=device)
    flops = ctypes.c_int64(0)
    out = {}
    for name, bf16, iters in (('mfma_f32_TFLOPs', 0, 1500), ('mfma_bf16_TFLOPs', 1, 3000)):
        def body(bf16=bf16, iters=iters):
            _lib.check(lib.pgv_probe_mfma(bf16, iters, sink.data_ptr(), ctypes.byref(flops), _stream()), "pgv_probe_mfma")
        body()
        torch.cuda.synchronize()
        ms = min(_time_graph(_graph_of(body), reps=5) for _ in range(2))
        out[name] = round(flops.value / (ms * 1e-3) / 1e12, 1)

    def read():
        _lib.check(lib.pgv_probe_read(_FLUSH.data_ptr(), _FLUSH.numel(), sink.data_ptr(), _stream()), "pgv_probe_read")
    read()
    torch.cuda.synchronize()
    ms = min(_time_graph(_graph_of(read), reps=5) for _ in range(2))
    out['hbm_read_GBs'] = round(_FLUSH.numel() * 4 / (ms * 1e-3) / 1e9, 1)
    _PEAKS = out
    return out


def launch_table(ae, B, device, frontend=None):
    """Every distinct kind of launch the train step issues, as (label, fn, algorithmic bytes, algorithmic flops):
    forward / input-gradient / weight-gradient of every conv block exactly as the step issues them, the BatchNorm /
    activation backward passes, the output-block criterion backward, the fc GEMMs, Adam, and (``--input audio``) the
    STFT->mel front-end.  Algorithmic bytes = every operand and result moved once."""
    from preset_gen_vae_amd import ops
    from preset_gen_vae_amd.model import layer as layer_mod
    table = []
    prev_bn = {False: False, True: False}   # does the producer block of the same stack end in a BatchNorm (folded here)?
    layers = layer_ops(ae)
    fused_bwd = True   # (layer.BN_BACKWARD_MODE 'fused'; bf16 operand mode: from B*H*W = 2^17 per channel on)
    sq_in_fwd = False  # the criterion rides in the output layer's forward kernel (set below where that kernel exists)
    for li, (name, (Cb, Cs, k, s, p, Hb, Wb), has_bn, is_up) in enumerate(layers):
        fold = prev_bn[is_up]
        prev_bn[is_up] = has_bn
        # the block below in the same stack (its BatchNorm + activation backward rides in this block's input gradient)
        lower = layers[li - 1] if li > 0 and layers[li - 1][3] == is_up else None
        # is this block's g_y the output gradient of a stride-2 ConvTranspose2d whose class sums the step keeps?
        upper = layers[li + 1] if li + 1 < len(layers) and layers[li + 1][3] == is_up else None
        geom = ops.ConvGeom(Cb, Cs, k, s, p, Hb, Wb)
        big = torch.randn(B, Cb, Hb, Wb, device=device)
        small = torch.randn(B, Cs, geom.Hs, geom.Ws, device=device)
        w = torch.randn(Cs, Cb, k, k, device=device) * 0.05
        gw = torch.empty_like(w)
        sc_b, sh_b = torch.ones(Cb, device=device), torch.zeros(Cb, device=device)
        sc_s, sh_s = torch.ones(Cs, device=device), torch.zeros(Cs, device=device)
        bias_b, bias_s = torch.zeros(Cb, device=device), torch.zeros(Cs, device=device)
        # (BatchNorm statistics as the step keeps them: zeroed partial copies per XCD for the large planes)
        big_plane = (Hb * Wb if is_up else geom.Hs * geom.Ws) >= layer_mod.PASSFREE_MIN_PLANE
        nsc = ops.CLS_COPIES if big_plane else 1
        stats_b = torch.zeros(nsc * 2 * Cb, device=device, dtype=torch.float64)
        stats_s = torch.zeros(nsc * 2 * Cs, device=device, dtype=torch.float64)
        out_s, out_b = torch.empty_like(small), torch.empty_like(big)
        flops = 2.0 * B * Cs * geom.Hs * geom.Ws * Cb * k * k
        nb, ns, nw = big.numel() * 4, small.numel() * 4, w.numel() * 4
        # the launches exactly as the train step issues them: the layer's own direction carries bias + activation
        # (+ BN statistics when the block has a BatchNorm, + the producer's folded BN), the opposite direction is the
        # input-gradient product - with the lower block's BatchNorm + activation backward in its epilogue
        # (pgv_bwd_fuse: the saved activation is one more operand read) when there is a lower block in the stack
        fwd_stats_s = stats_s if (has_bn and not is_up) else None
        fwd_stats_b = stats_b if (has_bn and is_up) else None

        if not fold:   # (first block of a stack, or a producer without BatchNorm: nothing to fold)
            sc_b = sh_b = sc_s = sh_s = None
        fuse = None
        lo = small if is_up else big                         # the lower block's output = this block's input
        big_n = ops.compute_dtype() == 'fp32' or B * lo.shape[2] * lo.shape[3] >= layer_mod.BF16_PASSFREE_MIN_N
        if lower is not None and fused_bwd and big_n and lo.shape[2] * lo.shape[3] >= layer_mod.PASSFREE_MIN_PLANE:
            Cl = lo.shape[1]
            a_lo = torch.randn_like(lo)
            coef = torch.cat([torch.ones(Cl, device=device), 0.01 * torch.randn(2 * Cl, device=device)])
            gb_lo = torch.zeros(ops.CLS_COPIES * Cl, device=device)   # (per-XCD partial copies: pgv_bwd_fuse.gbias_copies)
            # (class sums are kept when the lower block is itself a stride-2 ConvTranspose2d with a block below it)
            lower2 = layers[li - 2] if li > 1 and layers[li - 2][3] == is_up else None
            if lower2 is not None:   # ... whose own output plane is large enough for the pass-free backward
                lg = ops.ConvGeom(*lower[1])
                if lg.Hs * lg.Ws < layer_mod.PASSFREE_MIN_PLANE:
                    lower2 = None
            cls_lo = torch.zeros(ops.CLS_COPIES * 4 * Cl, device=device) if (is_up and lower2 is not None) else None
            fuse = (a_lo, coef, gb_lo, 1, 0.1, cls_lo, ops.CLS_COPIES)

        # the producer's train-mode BatchNorm is finalized in this block's forward kernel (ops.bn_src: statistics of a
        # zero-mean unit-variance input stand in), no launch of its own
        in_bn = None
        if fold:
            Ci, ni = lo.shape[1], B * lo.shape[2] * lo.shape[3]
            st_in = torch.cat([torch.zeros(Ci, dtype=torch.float64), torch.full((Ci,), float(ni), dtype=torch.float64)]).to(device)
            vec_in = [torch.empty(Ci, device=device) for _ in range(4)]
            in_bn = ops.bn_src(st_in, ni, torch.ones(Ci, device=device), torch.zeros(Ci, device=device), 1e-5, 0.1,
                               torch.zeros(Ci, device=device), torch.ones(Ci, device=device), None, *vec_in)
        # the BatchNorm-backward coefficients of the lower block ride in this block's weight-gradient call
        # (pgv_conv_wgrad_coef: partial-gradient reduce + border tap sums in one launch, then the coefficient kernel)
        coef_req = None
        if fuse is not None and lower[2]:
            Cl, gy_t = lo.shape[1], (big if is_up else small)
            m_ = s if is_up else 1
            one4 = [torch.ones(Cl, device=device) for _ in range(4)]
            coef_req = dict(lower_is_big=not is_up,
                            cls=torch.zeros(gy_t.shape[1] * m_ * m_ * (ops.CLS_COPIES if is_up else 1), device=device), w=w,
                            scale=one4[0], shift=one4[1], mean=one4[2], rstd=one4[3], n=lo.numel() // Cl,
                            coef=torch.empty(3 * Cl, device=device), ggamma=torch.empty(Cl, device=device),
                            gbeta=torch.empty(Cl, device=device),
                            scratch=torch.zeros(ops.coef_scratch(geom, not is_up), device=device, dtype=torch.float64))

        # this block's own bias gradient arrives as partial copies when the block above fused its backward
        # (pgv_bias_req: added up by the reduce launch of this weight gradient)
        bias_fin = None
        a_self = big if is_up else small
        if upper is not None and fused_bwd and (ops.compute_dtype() == 'fp32' or B * a_self.shape[2] * a_self.shape[3] >=
                                                layer_mod.BF16_PASSFREE_MIN_N) and \
                a_self.shape[2] * a_self.shape[3] >= layer_mod.PASSFREE_MIN_PLANE:
            Cself = a_self.shape[1]
            bias_fin = (torch.zeros(ops.CLS_COPIES * Cself, device=device), torch.zeros(Cself, device=device), False)

        def mk(kind, geom=geom, big=big, small=small, w=w, gw=gw, sc_b=sc_b, sh_b=sh_b, sc_s=sc_s, sh_s=sh_s,
               bias_b=bias_b, bias_s=bias_s, out_s=out_s, out_b=out_b, fs=fwd_stats_s, fb=fwd_stats_b, is_up=is_up,
               fuse=fuse, in_bn=in_bn, coef_req=coef_req, sck=nsc > 1, bias_fin=bias_fin,
               w_sh=ops.conv_weight_shadow(geom, w)):   # (bf16 operand mode, deep layers: as ConvStackFn passes it)
            if kind == 'conv_down':
                if is_up:
                    return lambda: ops.conv_down(geom, big, w, None, 0, 0.0, out=out_s, bwd_fuse=fuse, w_shadow=w_sh)
                if in_bn is not None:
                    return lambda: ops.conv_down(geom, big, w, bias_s, 1, 0.1, stats=fs, out=out_s, in_bn=in_bn,
                                                 prezeroed=fs is not None, stats_copies=sck and fs is not None, w_shadow=w_sh)
                return lambda: ops.conv_down(geom, big, w, bias_s, 1, 0.1, stats=fs, out=out_s, prezeroed=fs is not None,
                                             stats_copies=sck and fs is not None, w_shadow=w_sh)
            if kind == 'conv_up':
                if not is_up:
                    return lambda: ops.conv_up(geom, small, w, None, 0, 0.0, out=out_b, bwd_fuse=fuse, w_shadow=w_sh)
                if in_bn is not None:
                    return lambda: ops.conv_up(geom, small, w, bias_b, 1, 0.1, stats=fb, out=out_b, in_bn=in_bn,
                                               prezeroed=fb is not None, stats_copies=sck and fb is not None, w_shadow=w_sh)
                return lambda: ops.conv_up(geom, small, w, bias_b, 1, 0.1, stats=fb, out=out_b, prezeroed=fb is not None,
                                           stats_copies=sck and fb is not None, w_shadow=w_sh)
            return (lambda: ops.conv_wgrad(geom, big, small, gw, big_scale=sc_b, big_shift=sh_b, coef_req=coef_req,
                                           bias_finish=bias_fin)) \
                if not is_up else \
                (lambda: ops.conv_wgrad(geom, big, small, gw, small_scale=sc_s, small_shift=sh_s, coef_req=coef_req,
                                        bias_finish=bias_fin))

        for kind in ('conv_down', 'conv_up', 'conv_wgrad'):
            if name == 'enc1' and kind == 'conv_up':
                continue  # the first block needs no input gradient: this launch is not part of the train step
            is_dgrad = kind == ('conv_down' if is_up else 'conv_up')
            extra = (ns if is_up else nb) if (is_dgrad and fuse is not None) else 0   # the saved activation
            label = f"{kind}[{name}]" + ("+bn_act_bwd" if (is_dgrad and fuse is not None) else "")
            if kind == 'conv_wgrad' and coef_req is not None:
                label += "+bn_bwd_coef"
            if not is_dgrad and kind != 'conv_wgrad' and in_bn is not None:
                label += "+bn_finalize"
            if is_up and not is_dgrad and kind == 'conv_up' and big.shape[1] == 1 and li == len(layers) - 1:
                # the output layer of the step carries the reconstruction criterion in its epilogue where it has the kernel for
                # it (pgv_conv_up_sqerr: target in, gradient out, class sums / bias gradient / value as by-products)
                sq_t, sq_gb = torch.randn_like(big), torch.zeros(1, device=device)
                sq_cls = torch.zeros(ops.CLS_COPIES * 4, device=device)
                sq_fn = (lambda geom=geom, small=small, w=w, bias_b=bias_b, in_bn=in_bn, sc_s=sc_s, sh_s=sh_s:
                         ops.conv_up_sq(geom, small, w, bias_b, 2, 0.0, sq_t, 1.0 / sq_t.numel(), sq_gb, None, sq_cls,
                                        in_scale=None if in_bn is not None else sc_s,
                                        in_shift=None if in_bn is not None else sh_s, in_bn=in_bn))
                if sq_fn() is not None:
                    table.append((label + "+sqerr", sq_fn, nb + ns + nw + 2 * nb, flops))
                    sq_in_fwd = True
                    continue
            table.append((label, mk(kind), nb + ns + nw + extra, flops))
        a = big if is_up else small
        C = a.shape[1]
        big_na = ops.compute_dtype() == 'fp32' or B * a.shape[2] * a.shape[3] >= layer_mod.BF16_PASSFREE_MIN_N
        if has_bn and fused_bwd and big_na and upper is not None and a.shape[2] * a.shape[3] >= layer_mod.PASSFREE_MIN_PLANE:
            pass   # pass-free BatchNorm backward: its coefficients ride in the consumer block's weight-gradient entry
        elif has_bn:
            # top block of a stack (or bf16 operand mode): the reduce + apply passes over this block's output tensor
            g_o, g_y = torch.randn_like(a), torch.empty_like(a)
            mean, rstd, scale = torch.zeros(C, device=device), torch.ones(C, device=device), torch.ones(C, device=device)
            red = torch.zeros(2 * C, device=device, dtype=torch.float64)
            gbias, gga, gbe = (torch.zeros(C, device=device) for _ in range(3))
            if ops.bn_act_bwd_fusable(B, C, a.shape[2] * a.shape[3]):   # (the deepest blocks: one launch, as in the step)
                table.append((f"bn_act_bwd_fused[{name}]",
                              (lambda g_o=g_o, a=a, scale=scale, mean=mean, rstd=rstd, g_y=g_y, gbias=gbias, gga=gga, gbe=gbe:
                               ops.bn_act_bwd_fused(g_o, a, scale, mean, rstd, 1, 0.1, g_y, gbias, ggamma=gga, gbeta=gbe)),
                              3 * a.numel() * 4, 0.0))
                continue
            table.append((f"act_bn_bwd[{name}]",
                          (lambda g_o=g_o, a=a, scale=scale, mean=mean, rstd=rstd, red=red, g_y=g_y, gbias=gbias, gga=gga,
                           gbe=gbe: ops.act_bn_bwd(g_o, a, scale, mean, rstd, red, 1, 0.1, g_y, gbias, ggamma=gga,
                                                   gbeta=gbe)), 3 * a.numel() * 4, 0.0))
            table.append((f"bn_bwd_reduce[{name}]",
                          (lambda g_o=g_o, a=a, mean=mean, rstd=rstd, red=red: ops.bn_bwd_reduce(g_o, a, mean, rstd, red)),
                          2 * a.numel() * 4, 0.0))
    # output block: criterion + Hardtanh backward in one pass (pgv_sqerr_act_bwd)
    xo, xt, gy = (torch.randn(B, 1, 257, 347, device=device) for _ in range(3))
    gl, gb1 = torch.ones((), device=device), torch.zeros(1, device=device)
    cls8 = torch.zeros(ops.CLS_COPIES * 4, device=device) if fused_bwd else None   # (class sums of g_y for the block below, as in the step)
    if not sq_in_fwd:
        table.append(("sqerr_act_bwd[dec8]", lambda: ops.sqerr_act_bwd(xo, xt, gl, 1.0 / xo.numel(), 2, 0.0, gy, gb1,
                                                                        prezeroed=True, cls=cls8),
                      3 * xo.numel() * 4, 0.0))
    # fc layers (encoder.mlp.1 / decoder.mlp.0): forward, input gradient, weight gradient
    lin_e, lin_d = ae.encoder.mlp[1], ae.decoder.mlp[0]
    for tag, lin in (('enc_fc', lin_e), ('dec_fc', lin_d)):
        N, K = lin.weight.shape
        xin, gyl = torch.randn(B, K, device=device), torch.randn(B, N, device=device)
        wl, bl, gwl = lin.weight.detach(), lin.bias.detach(), torch.empty_like(lin.weight)
        byt, fl = (B * K + B * N + N * K) * 4, 2.0 * B * N * K
        table.append((f"linear_fwd[{tag}]", lambda xin=xin, wl=wl, bl=bl: ops.linear_fwd(xin, wl, bl), byt, fl))
        table.append((f"linear_dgrad[{tag}]", lambda gyl=gyl, wl=wl: ops.linear_dgrad(gyl, wl), byt, fl))
        table.append((f"linear_wgrad[{tag}]", lambda gyl=gyl, xin=xin, gwl=gwl: ops.linear_wgrad(gyl, xin, gwl), byt, fl))
    # fc Dropout without a stored mask (forward draws and applies, backward regenerates), the encoder's with its last
    # conv block's BatchNorm folded in, the decoder's backward with the Linear bias gradient on the way
    from preset_gen_vae_amd.rng import DeviceRNG
    rng = DeviceRNG(device, seed=1)
    K = lin_e.weight.shape[1]
    xd = torch.randn(B, K, device=device)
    _, saved = ops.dropout_fwd(rng.state, 2, 0.3, xd)
    cs = torch.zeros(K, device=device)
    enc_last = [l for l in layers if not l[3]][-1]
    if enc_last[2]:
        g_l = ops.ConvGeom(*enc_last[1])
        Ce, ne = enc_last[1][1], B * g_l.Hs * g_l.Ws
        xe = torch.randn(B, Ce, g_l.Hs, g_l.Ws, device=device)
        st_e = torch.cat([torch.zeros(Ce, dtype=torch.float64), torch.full((Ce,), float(ne), dtype=torch.float64)]).to(device)
        bn_e = ops.bn_src(st_e, ne, torch.ones(Ce, device=device), torch.zeros(Ce, device=device), 1e-5, 0.1,
                          torch.zeros(Ce, device=device), torch.ones(Ce, device=device), None,
                          *[torch.empty(Ce, device=device) for _ in range(4)])
        table.append(("dropout_fwd[enc_fc]+bn_finalize", lambda: ops.dropout_fwd(rng.state, 2, 0.3, xe, in_bn=bn_e),
                      2 * xe.numel() * 4, 0.0))
    table.append(("dropout_fwd[dec_fc]", lambda: ops.dropout_fwd(rng.state, 3, 0.3, xd), 2 * xd.numel() * 4, 0.0))
    table.append(("dropout_bwd[enc_fc]", lambda: ops.dropout_bwd(saved, 2, 0.3, xd), 2 * xd.numel() * 4, 0.0))
    table.append(("dropout_bwd+colsum[dec_fc]", lambda: ops.dropout_bwd(saved, 3, 0.3, xd, colsum=cs, prezeroed=True),
                  2 * xd.numel() * 4, 0.0))
    # fused Adam over the flat parameter buffer: read p, g, m, v; write p, m, v
    n = sum(p.numel() for p in ae.parameters())
    fp, fg, fm, fv = (torch.zeros(n, device=device) for _ in range(4))
    hyper = torch.tensor([2e-4, 0.1, 0.001, 1.0], device=device)
    table.append(("adam", lambda: ops.adam_step(fp, fg, fm, fv, hyper, 0.9, 0.999, 1e-8, 1e-4), 7 * n * 4, 0.0))
    if frontend is not None:
        wav = 0.3 * torch.randn(B, 88576, device=device)
        xo2 = torch.empty(B, 1, 257, 347, device=device)
        table.append(("stft_mel", lambda: frontend.batch(wav, out=xo2), B * (88576 + 257 * 347) * 4, 0.0))
    return table


def measure_roofline(ae, B, device, step_ms, matrix_peak=F32_MATRIX_PEAK_TFLOPS, frontend=None, traffic_file=None,
                     traffic_ok=False):
    """Time every kind of launch of the step standalone at the bench shapes and report the one FURTHEST BELOW ITS OWN
    ROOFLINE among the launches that take at least 2 % of the step (the kernel to fix next); the table of all launches
    goes to gpurun_out/bench_kernel_table.json."""
    from preset_gen_vae_amd import ops
    rows = []
    # six-instruction products: a conv launch of a layer with more than one input channel issues 6 bf16 instructions per
    # fp32 product, so ITS matrix roofline is the dense bf16 peak / 6 (417 TFLOP/s of fp32 products) - not the fp32
    # instruction's peak, which those kernels do not use
    split_layers = set()
    if ops.compute_dtype() == 'fp32' and ops.fp32_products() != 'native':
        split_layers = {name for name, (Cb, Cs, k, s, p, Hb, Wb), _, _ in layer_ops(ae) if Cb > 1}
    for label, fn, bytes_, flops in launch_table(ae, B, device, frontend):
        # median of three 5-launch averages: one disturbed replay (another tenant of the box, a clock ramp) would otherwise
        # push a launch over the 2 % line AND to the bottom of the fractions at once (seen: conv_wgrad[dec8] 0.43 -> 0.25)
        ms = sorted(time_kernel(fn, iters=5) for _ in range(3))[1]
        lname = label[label.index('[') + 1:label.index(']')] if '[' in label else ''
        row_peak = BF16_MATRIX_PEAK_TFLOPS / 6.0 if (label.startswith('conv_') and lname in split_layers) else matrix_peak
        t_hbm, t_mfma = bytes_ / (HBM_PEAK_GBS * 1e9), flops / (row_peak * 1e12)
        bound = 'hbm' if t_hbm >= t_mfma else 'mfma'
        frac = max(t_hbm, t_mfma) * 1e3 / ms if ms > 0 else 0.0
        rows.append({'launch': label, 'ms': ms, 'flops': flops, 'bytes': bytes_, 'bound': bound, 'frac': frac,
                     'share_of_step': ms / step_ms, 'matrix_peak': row_peak})
    # (launch-latency-sized helpers with no algorithmic bytes / flops of their own have no roofline to stand against)
    priced = [r for r in rows if r['bytes'] > 0 or r['flops'] > 0]
    cands = [r for r in priced if r['share_of_step'] >= 0.02] or priced
    worst = min(cands, key=lambda r: r['frac'])
    if worst['bound'] == 'hbm':
        achieved, peak, unit = worst['bytes'] / (worst['ms'] * 1e-3) / 1e9, HBM_PEAK_GBS, 'GB/s'
    else:
        achieved, peak, unit = worst['flops'] / (worst['ms'] * 1e-3) / 1e12, round(worst['matrix_peak'], 1), 'TFLOP/s'
    # HBM bytes per launch from the committed rocprofv3 PMC passes of THIS kernel generation and THIS configuration at batch
    # 256 (the same labels on another model / operand mode are other kernels or other fusions)
    traffic = None
    try:
        # (the front-end launch has a PMC file of its own: profiles/pmc_frontend.py)
        tf = os.path.join(ROOT, 'profiles', 'r6_traffic_frontend.json') if worst['launch'] == 'stft_mel' else \
            (traffic_file or os.path.join(ROOT, 'profiles', 'r6_traffic.json'))
        with open(tf) as f:
            entry = json.load(f).get(worst['launch'])
        if entry and B == 256 and traffic_ok:
            traffic = entry['hbm_bytes_per_launch']
    except (OSError, ValueError):
        pass
    conv = [r for r in rows if r['launch'].startswith('conv_')]
    mp = measured_peaks(device)
    mpeak = mp['hbm_read_GBs'] if worst['bound'] == 'hbm' else \
        (mp['mfma_bf16_TFLOPs'] / 6.0 if worst['matrix_peak'] != matrix_peak else
         (mp['mfma_bf16_TFLOPs'] if matrix_peak == BF16_MATRIX_PEAK_TFLOPS else mp['mfma_f32_TFLOPs']))
    # the same sum of per-launch rooflines, priced against what this box sustains instead of the data-sheet figures
    mpf = mp['mfma_bf16_TFLOPs'] if matrix_peak == BF16_MATRIX_PEAK_TFLOPS else mp['mfma_f32_TFLOPs']

    def row_mpf(r):   # (a six-instruction launch: the measured bf16 rate / 6)
        return mp['mfma_bf16_TFLOPs'] / 6.0 if r['matrix_peak'] not in (matrix_peak,) else mpf
    conv_meas = sum(max(r['bytes'] / (mp['hbm_read_GBs'] * 1e9), r['flops'] / (row_mpf(r) * 1e12)) for r in conv) * 1e3
    roof = {'bound': worst['bound'], 'achieved': round(achieved, 3), 'peak': peak, 'unit': unit,
            'frac': round(achieved / peak, 5), 'measured_peak': mpeak, 'frac_of_measured_peak': round(achieved / mpeak, 5),
            'measured_peaks': mp, 'conv_launches_sum_measured_roofline_ms': round(conv_meas, 4),
            'traffic': traffic, 'kernel': worst['launch'],
            'kernel_ms': round(worst['ms'], 4), 'algorithmic_bytes': worst['bytes'],
            'algorithmic_flops': worst['flops'],
            'selection': 'lowest roofline fraction among launches >= 2 % of the step',
            'conv_launches_sum_roofline_ms': round(sum(r['ms'] * r['frac'] for r in conv), 4),
            'conv_launches_sum_ms': round(sum(r['ms'] for r in conv), 4)}
    # the WHOLE step against its roofline: sum over every launch of the table of its own bound - max(algorithmic
    # bytes / 8 TB/s, flops / matrix peak of the instruction it uses) - divided by the measured time of a step; 'hbm_frac' =
    # the step's algorithmic bytes / step time against 8 TB/s alone
    t_roof = sum(r['ms'] * r['frac'] for r in priced)
    all_bytes = sum(r['bytes'] for r in priced)
    roof['step'] = {'sum_of_launch_rooflines_ms': round(t_roof, 4), 'ms_per_step': round(step_ms, 4),
                    'frac': round(t_roof / step_ms, 5), 'algorithmic_bytes': int(all_bytes),
                    'hbm_frac': round(all_bytes / (step_ms * 1e-3) / (HBM_PEAK_GBS * 1e9), 5),
                    'sum_of_launch_times_ms': round(sum(r['ms'] for r in rows), 4)}
    return roof, rows


def _cpu_model():
    try:
        with open('/proc/cpuinfo') as f:
            for ln in f:
                if ln.lower().startswith('model name'):
                    return ln.split(':', 1)[1].strip()
    except OSError:
        pass
    return None


def _frontend_cpu_worker(job):
    """One worker process of the all-cores CPU front-end baseline: per-item loop for ``seconds``, returns the item count."""
    seed, seconds = job
    try:   # one BLAS / OpenMP thread per worker process: the workers ARE the parallelism (256 processes x a threaded BLAS
        #    each gave 148 waveforms/s on a 256-thread host against 121 on one core)
        from threadpoolctl import threadpool_limits
        threadpool_limits(limits=1)
    except ImportError:
        pass
    torch.set_num_threads(1)
    from oracle import audio_oracle as ao   # CPU baseline leg only
    from preset_gen_vae_amd.utils.synthetic import fm_voice
    waves = [fm_voice(idx=seed * 4 + i) for i in range(4)]
    ao.minmax_normalize(ao.mel_spectrogram_db(waves[0], dtype=np.float32), -120.0, 0.0)   # (warm-up: plans, basis)
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds:
        ao.minmax_normalize(ao.mel_spectrogram_db(waves[n % 4], dtype=np.float32), -120.0, 0.0)
        n += 1
    return n, time.perf_counter() - t0


_FRONTEND_ALL_CORES = None


def frontend_cpu_all_cores(seconds=5.0):
    """BASELINE.md section 4 asks for the front-end's CPU baseline on all host cores as well as on one: the reference's
    DataLoader runs its per-item ``__getitem__`` loop (data/abstractbasedataset.py:126-134) in worker processes, so this leg
    does the same - one forked worker per core, each looping over items for ``seconds``.  It runs at the very start of
    bench.py, BEFORE this process makes its first GPU call (a process that has initialised HIP must not fork workers)."""
    global _FRONTEND_ALL_CORES
    import multiprocessing as mp
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 128))   # (one worker per core up to 128: beyond that the fork + import cost eats the sample)
    try:
        with mp.get_context('fork').Pool(cores) as pool:
            res = pool.map(_frontend_cpu_worker, [(i, seconds) for i in range(cores)])
        _FRONTEND_ALL_CORES = {'value': round(sum(n / t for n, t in res), 2), 'unit': 'waveforms/s', 'cores': cores,
                               'kind': 'port',
                               'sample': f'{sum(n for n, _ in res)} items over {cores} worker processes x {seconds:.0f} s, numpy '
                                         f'float32, per-item loop as data/abstractbasedataset.py:126-134'}
    except (OSError, ValueError) as e:   # (no fork / no semaphores in a sandbox: report why instead of a number)
        _FRONTEND_ALL_CORES = {'value': None, 'unit': 'waveforms/s', 'cores': cores, 'kind': 'port', 'sample': f'not measured: {e}'}
    return _FRONTEND_ALL_CORES


def frontend_figures(device, B=256, iters=10, cpu_seconds=6.0):
    """the STFT -> mel -> dB -> min-max front-end on its own - GPU waveforms/s (HIP events over replays
    of the batched kernel, inputs resident) against its HBM roofline (0.711 MB per spectrogram), and the CPU restatement
    (oracle/audio_oracle.py, numpy) in the reference's per-item loop (data/abstractbasedataset.py:126-134), bounded."""
    from preset_gen_vae_amd.utils.audio import MelSpectrogram
    from preset_gen_vae_amd.utils.synthetic import fm_voice
    waves = np.stack([fm_voice(idx=i) for i in range(16)])
    wav = torch.tensor(np.tile(waves, (B // 16 + 1, 1))[:B], device=device)
    mel = MelSpectrogram(1024, 256, -120.0, 257, 22050, device=device)
    mel.set_minmax_normalization(-120.0, 0.0)
    out = mel.batch(wav)
    ms = time_kernel(lambda: mel.batch(wav, out=out), iters=5)
    per = wav.shape[1] * 4 + 257 * out.shape[-1] * 4
    from oracle import audio_oracle as ao   # CPU baseline leg only
    t0, n = time.perf_counter(), 0
    while n < 4 or (time.perf_counter() - t0 < cpu_seconds and n < 64):
        ao.minmax_normalize(ao.mel_spectrogram_db(waves[n % 16], dtype=np.float32), -120.0, 0.0)
        n += 1
    cpu = n / (time.perf_counter() - t0)
    gbs = per * B / (ms * 1e-3) / 1e9
    return {'metric': 'waveforms/sec STFT->mel->dB front-end (88576 samples -> 1x257x347)',
            'value': round(B / ms * 1e3, 1), 'unit': 'waveforms/s', 'batch': B, 'ms_per_batch': round(ms, 4),
            'roofline': {'bound': 'hbm', 'achieved': round(gbs, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': round(gbs / HBM_PEAK_GBS, 5)},
            'cpu_baseline': {'value': round(cpu, 2), 'unit': 'waveforms/s', 'cores': 1, 'kind': 'port',
                             'sample': f'{n} items, numpy float32, per-item loop as data/abstractbasedataset.py:126-134'},
            'cpu_baseline_all_cores': _FRONTEND_ALL_CORES}


def h2d_inclusive(step, x, steps=10):
    """The PCIe-inclusive rate (never `value`): every step starts with a pinned host -> device copy of its minibatch
    into the captured step's input buffer, not overlapped with the previous step."""
    host = x.detach().cpu().pin_memory()
    dst = step.static_input if step.static_input is not None else x
    for _ in range(2):
        dst.copy_(host, non_blocking=True)
        step.step(dst)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        dst.copy_(host, non_blocking=True)
        step.step(dst)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    res = {'value': round(x.shape[0] / dt, 2), 'unit': 'spectrograms/s', 'ms_per_step': round(dt * 1e3, 4),
           'note': 'pinned host -> device copy of the minibatch (91 MB) in front of every step, not overlapped'}
    if step.static_input is not None and hasattr(step, 'prefetch_input'):
        # the loader-shaped form: the copy of minibatch n + 1 runs on a copy stream while step n replays
        # (VAETrainStep.prefetch_input / step_prefetched: a second device buffer + one device-to-device copy per step)
        step.prefetch_input(host)
        for _ in range(2):
            step.step_prefetched()
            step.prefetch_input(host)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step.step_prefetched()
            step.prefetch_input(host)
        torch.cuda.synchronize()
        dto = (time.perf_counter() - t0) / steps
        step.step_prefetched()
        torch.cuda.synchronize()
        res['overlapped'] = {'value': round(x.shape[0] / dto, 2), 'ms_per_step': round(dto * 1e3, 4),
                             'note': 'the copy of the next minibatch overlaps the running step (copy stream + staging buffer)'}
    return res


def cpu_baseline(arch, dim_z, B, max_seconds=25.0):
    """The oracle's full train step (torch CPU ops, fp32) on the host cores: a bounded sample of the same workload.
    torch's default (one thread per logical core: 128 on the GPU boxes) oversubscribes a batch-16 step, so a few thread
    counts are probed with one step each and the timed sample runs on the fastest - `cores` reports that count."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from helpers import param_shapes, synth_input, synth_vec
    from oracle import vae_oracle as vo
    sd = vo.closed_form_state_dict(param_shapes(arch, dim_z, False), seed=1234, dtype=torch.float32)
    x = synth_input(B, dtype=torch.float32)
    eps = synth_vec((B, dim_z), 1.2345, 0.4, dtype=torch.float32)
    default_threads = torch.get_num_threads()
    t_start = time.perf_counter()

    def one(state, i):
        t0 = time.perf_counter()
        r = vo.train_step(sd, x, arch, dim_z, eps, None, None, adam_state=state, step=i + 1)
        return r, time.perf_counter() - t0

    best_threads, best_dt = default_threads, None
    for n in sorted({min(default_threads, c) for c in (8, 16, 32, 64, default_threads)}):
        torch.set_num_threads(n)
        one(None, 0)                       # warm the thread pool / allocator at this width
        _, dt = one(None, 0)
        if best_dt is None or dt < best_dt:
            best_threads, best_dt = n, dt
        if time.perf_counter() - t_start > 0.5 * max_seconds:
            break
    torch.set_num_threads(best_threads)
    state, times, dt = None, [], best_dt
    for i in range(13):
        r, dt = one(state, i)
        sd, state = r['new_sd'], r['adam_state']
        if i >= 3:
            times.append(dt)
        if time.perf_counter() - t_start > max_seconds and len(times) >= 2:
            break
    # one thread (SURVEY section 8d asks for it next to the all-cores figure): a single step, bounded
    torch.set_num_threads(1)
    one_thread = None
    if time.perf_counter() - t_start < 1.6 * max_seconds:
        _, dt1 = one(state, 0)
        one_thread = round(B / dt1, 2)
    torch.set_num_threads(default_threads)
    med = float(np.median(times)) if times else dt
    return {'value': round(B / med, 2), 'unit': 'spectrograms/s', 'cores': best_threads, 'kind': 'port',
            'host_logical_cpus': os.cpu_count(), 'host_cpu_model': _cpu_model(), 'one_thread_value': one_thread,
            'sample': f'{len(times)} timed train steps (after 3 warm-up) of {arch} dz={dim_z} fp32 at batch {B} '
                      f'on torch CPU ops with {best_threads} threads (fastest of 8..{default_threads}), '
                      f'median {med * 1e3:.1f} ms/step'}


def run_workload(args, rank, world, device, with_roofline, with_cpu):
    """Build the model of ``args``, run ``args.warmup`` untimed + ``args.steps`` timed steps (barrier + synchronize on
    both sides, MAX over ranks) and return the JSON line of that workload as a dict (rank 0; None elsewhere)."""
    import torch.distributed as dist
    from preset_gen_vae_amd import config, ops, parallel
    from preset_gen_vae_amd.model import build as mbuild
    from preset_gen_vae_amd.train_step import VAETrainStep
    ops.set_compute_dtype(args.dtype)
    if getattr(args, 'gemm_variant', 0):
        from preset_gen_vae_amd import _lib
        _lib.load().pgv_dbg_set_gemm_variant(int(args.gemm_variant))
    # (None = the library default: the headline line times exactly what the training path runs by default)
    ops.set_fp32_products(args.fp32_products if args.dtype == 'fp32' else 'native')
    mc, tc = copy.copy(config.model), copy.copy(config.train)
    tc.fp32_products = ops.fp32_products() if args.dtype == 'fp32' else None   # (the config field build_ae_model reads)
    mc.encoder_architecture, mc.dim_z = args.arch, args.dim_z
    tc.minibatch_size = args.batch
    mc.input_tensor_size = (args.batch, 1, 257, 347)
    # reference default (config.py:92): 'bn' -> BatchNorm1d on the encoder output (build.py:25)
    tc.latent_flow_input_regularization = args.latent_reg
    torch.manual_seed(1234)            # identical replicas on every rank
    _, _, ae = mbuild.build_ae_model(mc, tc)
    ae = ae.to(device).train()
    torch.manual_seed(1234 + rank)     # per-rank eps / dropout streams
    x = synth_spectrograms(args.batch, device, seed=rank)

    # N = 1: one hipGraph per step.  N > 1: see --dist-mode
    dist_on = world > 1 or getattr(args, 'force_dist', False)
    use_graph = (not args.no_graph) and (not dist_on or args.dist_mode != 'eager')
    sync = None
    if dist_on:
        sync = (lambda flat: parallel.GradAllReduce(flat, n_buckets=args.buckets))
    launch_fallback = None

    frontend = None
    if args.input == "audio":
        from preset_gen_vae_amd.utils.audio import MelSpectrogram
        frontend = MelSpectrogram(1024, 256, -120.0, 257, 22050)
        frontend.set_minmax_normalization(-120.0, 3.5)
        gen = torch.Generator(device=device)
        gen.manual_seed(1234 + rank)
        wav = 0.3 * torch.randn(args.batch, 88576, device=device, generator=gen)   # 173 render buffers of 512 samples
        x = frontend.batch(wav)

    # (--input audio: the front-end is the step's input producer - captured with it, it writes the step's input buffer from the
    # resident waveforms as the first launch of every replay)
    producer = (lambda buf: frontend.batch(wav, out=buf)) if frontend is not None else None

    def one_step(xin):
        return step.step(xin)

    # N > 1: the captured launch modes have run over gloo and over a 1-rank RCCL communicator, never over RCCL on several
    # GPUs (no SCALE record exists).  So the launch mode is chosen by a LADDER - the requested mode, then two-graph (the same
    # kernels, one eager exchange between two replays), then eager - and every rung is tried with at least one protected
    # warm-up step (also with --warmup 0: the capture never happens inside the timed loop), after which the ranks AGREE on
    # the outcome (MAX all-reduce of a failure flag) before any of them goes on: a rank whose capture failed never leaves
    # the others waiting in a collective of a mode it has abandoned.
    ladder = [args.dist_mode] + [m for m in ('two-graph', 'eager') if m != args.dist_mode] if (dist_on and use_graph) \
        else [args.dist_mode if dist_on else 'single']
    step, failures = None, []
    for mode in ladder:
        graph = use_graph and mode != 'eager'
        err = None
        try:
            step = VAETrainStep(ae, lr=tc.initial_learning_rate, betas=tc.adam_betas, weight_decay=tc.weight_decay,
                                beta=tc.beta, normalize_losses=tc.normalize_losses, grad_sync=sync, use_graph=graph,
                                graph_buckets=mode == 'bucket-graphs', input_producer=producer)
            for _ in range(max(1, args.warmup) if dist_on else args.warmup):
                out = one_step(x)
            torch.cuda.synchronize()
        except RuntimeError as e:
            if not (dist_on and graph):
                raise
            err = str(e).splitlines()[0][:160]
        failed = torch.tensor([1.0 if err else 0.0], device=device)
        if world > 1:
            dist.all_reduce(failed, op=dist.ReduceOp.MAX)
        if failed.item() == 0.0:
            args = copy.copy(args)
            args.dist_mode, use_graph = (mode if dist_on else args.dist_mode), graph
            break
        failures.append(f"{mode} failed ({err or 'on another rank'})")
        print(f"[bench rank {rank}] {failures[-1]}; trying the next launch mode", file=sys.stderr, flush=True)
        if step is not None and step.grad_sync is not None:
            step.grad_sync.uninstall()
        step = None
        torch.cuda.synchronize()
    if step is None:
        raise SystemExit("bench.py: no launch mode of the N > 1 step worked: " + "; ".join(failures))
    if failures:
        launch_fallback = "; ".join(failures) + f"; measured with {args.dist_mode} launches instead"
    if step.grad_sync is not None and dist_on:
        step.grad_sync.time_buckets(True)      # event pairs around every bucket's collective (reported per bucket)
    if step.static_input is not None:
        # the minibatch lives in the captured step's input buffer (where the on-GPU front-end / the H2D copy of a real
        # loader would put it): inputs are resident in HBM when the timed region starts, no device-to-device copy
        x = step.static_input
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    n_coll0 = step.grad_sync.n_collectives if step.grad_sync is not None else 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = one_step(x)
    torch.cuda.synchronize()
    n_coll = (step.grad_sync.n_collectives - n_coll0) if step.grad_sync is not None else 0
    bucket_ms = step.grad_sync.bucket_times_ms() if (step.grad_sync is not None and dist_on) else None
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    loss = out['total'].item()
    assert np.isfinite(loss), "non-finite loss in the timed region"
    ms = elapsed / args.steps * 1e3

    roof, table, cpu, h2d = None, None, None, None
    if rank == 0 and world == 1 and with_cpu and frontend is None:
        h2d = h2d_inclusive(step, x)
    if rank == 0 and with_roofline:
        # the PMC passes exist for three configurations (profiles/README.md); other configurations report traffic null
        prod = ops.fp32_products() if args.dtype == 'fp32' else 'native'
        tfile = {('speccnn4l1_bn', 'fp32', 64, 'bf16x6'): 'r6_traffic.json',
                 ('speccnn8l1_bn', 'fp32', 64, 'bf16x6'): 'r6_traffic_8l.json',
                 ('speccnn4l1_bn', 'fp32', 64, 'native'): 'r6_traffic_native.json',
                 ('speccnn8l1_bn', 'fp32', 64, 'native'): 'r6_traffic_8l_native.json',
                 ('speccnn8l1_bn', 'bf16', 512, 'native'): 'r6_traffic_8l_bf16.json'}.get((args.arch, args.dtype, args.dim_z, prod))
        roof, table = measure_roofline(ae, args.batch, device, ms, BF16_MATRIX_PEAK_TFLOPS if args.dtype == 'bf16'
                                       else F32_MATRIX_PEAK_TFLOPS, frontend=frontend,
                                       traffic_file=os.path.join(ROOT, 'profiles', tfile) if tfile else None,
                                       traffic_ok=tfile is not None)
    if rank == 0 and world == 1 and with_cpu:
        cpu = cpu_baseline(args.arch, args.dim_z, args.cpu_batch)
    if world > 1:
        dist.barrier()
    line = None
    if rank == 0:
        value = args.batch * world * args.steps / elapsed
        launch = "eager"
        if use_graph:
            launch = "hipGraph"
            if dist_on and args.dist_mode == 'two-graph':
                launch = "2 hipGraphs + eager all-reduce"
            elif dist_on:
                launch = (f"{sum(1 for g_, _ in step._bucket_graphs if g_ is not None) + 1} hipGraphs cut at the gradient buckets, all-reduce of a bucket "
                          f"launched between two replays (overlaps the rest of backward)")
        cfg = {"workload": f"{args.arch} conv-VAE dz={args.dim_z} {args.dtype} full train step "
                           f"(fwd, MSE+0.2*KL, bwd, Adam), batch {args.batch}/GPU, 1x257x347 log-mel"
                           + (" computed on the GPU from raw audio [B, 88576] inside the timed step"
                              if args.input == "audio" else ""),
               "global_batch": args.batch * world, "parallelism": f"dp{world}", "launch": launch,
               "latent_flow_input_regularization": args.latent_reg, "final_loss": round(loss, 6)}
        if args.dtype == 'fp32':
            cfg["fp32_products"] = (
                "bf16x6: the k4 s2 layers from 129x174 down to 5x7 and the 1x1 layers (forward, fused input gradient, weight "
                "gradient) evaluate every fp32 product as six v_mfma_f32_16x16x32_bf16 on exact "
                "three-way operand splits with fp32 accumulation (error against float64 <= the native instruction's: "
                "tests/test_gpu_kernels.py); the 1-channel 5x5 end layers and the fc GEMMs use the native fp32 instruction"
                if ops.fp32_products() != 'native' else "native: v_mfma_f32_16x16x4_f32 / fp32 FMA everywhere")
        if launch_fallback:
            cfg["launch_fallback"] = launch_fallback
            cfg["launch"] = launch + " - FALLBACK: " + launch_fallback
        if dist_on:
            cfg["rccl_ranks"] = dist.get_world_size()
            cfg["backend"] = dist.get_backend()
            cfg["grad_buckets_bytes"] = [4 * (hi - lo) for lo, hi in step.grad_sync.ranges]
            cfg["collective_launches_per_step"] = n_coll / max(1, args.steps)
            cfg["collective_ms_per_bucket"] = bucket_ms      # (events on the communication stream, timed steps only)
            cfg["multi_gpu_status"] = ("measured on %d ranks in this run" % world) if world > 1 else \
                "UNMEASURED on more than one GPU: the N > 1 launch path run over a 1-rank communicator (--force-dist)"
        line = {
            "metric": "spectrograms/sec per VAE train step (batch 256, 1x257x347)", "value": round(value, 2),
            "unit": "spectrograms/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if args.dtype == 'bf16' else "f32", "data": "synthetic", "config": cfg,
            "roofline": roof, "cpu_baseline": cpu,
        }
        if h2d is not None:
            line["h2d_inclusive"] = h2d
        if table is not None:
            line["_table"] = table
    if step.grad_sync is not None:
        step.grad_sync.uninstall()
    del step, ae, x
    ops.set_compute_dtype('fp32')
    ops.set_fp32_products(None)
    torch.cuda.empty_cache()
    return line


# the other BASELINE.json configurations measured inside the same invocation (N = 1): fewer steps, live roofline each
EXTRA_CONFIGS = [
    ("configs[1] with the native fp32 matrix instruction everywhere (ops.set_fp32_products('native'))",
     dict(fp32_products='native')),
    ("configs[1] on the reference-exact 8-layer stack", dict(arch='speccnn8l1_bn')),
    ("configs[1] on the 8-layer stack with the native fp32 matrix instruction everywhere",
     dict(arch='speccnn8l1_bn', fp32_products='native')),
    ("configs[2]: 8-layer z=512 bf16", dict(arch='speccnn8l1_bn', dim_z=512, dtype='bf16')),
    ("configs[4]: raw-audio minibatch, fused STFT->mel front-end", dict(input='audio')),
]


def self_launch(n):
    """``python bench.py --gpus N`` without a launcher: start the N ranks ourselves, as CHILD processes of
    ``torch.distributed.run`` (one per GPU, rendezvous on 127.0.0.1, a free port), and hand their exit status back.
    This process has made no GPU call yet (``import torch`` does not initialise HIP) and makes none afterwards; it never
    replaces itself with another program."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # dmabuf IPC: RCCL's intra-node transport on this driver
    env.setdefault('OMP_NUM_THREADS', '8')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    rank = int(os.environ.get('RANK') or 0)
    local_rank = int(os.environ.get('LOCAL_RANK') or 0)
    world = int(os.environ.get('WORLD_SIZE') or 1)
    if args.gpus > 1 and not os.environ.get('WORLD_SIZE'):   # (unset or empty: no launcher above us)
        raise SystemExit(self_launch(args.gpus))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run "
                         f"--nproc-per-node {args.gpus}, or without a launcher (bench.py then starts its own ranks)")
    if world == 1 and not args.no_cpu_baseline and not args.force_dist:
        frontend_cpu_all_cores()   # (forks worker processes: before the first GPU call of this process)
    import torch.distributed as dist
    from preset_gen_vae_amd import _lib
    _lib.load()   # fails loudly if the HIP library is missing
    assert torch.cuda.is_available(), "bench.py needs a ROCm GPU"
    # one rank per GPU; PGV_DIST_BACKEND=gloo + a single visible GPU lets the N>1 code path be exercised on a 1-GPU box
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device('cuda', dev_index)
    if world > 1:
        dist.init_process_group(os.environ.get('PGV_DIST_BACKEND', 'nccl'), rank=rank, world_size=world)
    elif args.force_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        dist.init_process_group(os.environ.get('PGV_DIST_BACKEND', 'nccl'), rank=0, world_size=1)

    line = run_workload(args, rank, world, device, not args.no_roofline, not args.no_cpu_baseline)
    extras = []
    if world == 1 and not args.no_extra and not args.force_dist:
        for label, over in EXTRA_CONFIGS:
            a2 = copy.copy(args)
            for k, v in over.items():
                setattr(a2, k, v)
            # (the same number of steps as the headline line: the first replays of a captured step run 5 - 10 % slower than
            # the following ones, 10 steps after 3 warm-ups read 2.5 - 3 % above 30 after 5 - 1.853 / 1.813 ms, same box)
            a2.steps, a2.warmup = args.steps, args.warmup
            e = run_workload(a2, rank, world, device, not args.no_roofline, False)
            e.pop("_table", None)
            e.pop("cpu_baseline", None)
            e["label"] = label
            extras.append(e)
    if rank == 0:
        table = line.pop("_table", None)
        if extras:
            line["extra"] = extras
        if world == 1 and not args.no_cpu_baseline and not args.force_dist:
            line["frontend"] = frontend_figures(device)
        if table is not None:
            os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
            with open(os.path.join(ROOT, 'gpurun_out', 'bench_kernel_table.json'), 'w') as f:
                json.dump(table, f, indent=1)
        print(json.dumps(line), flush=True)
    if world > 1 or args.force_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
