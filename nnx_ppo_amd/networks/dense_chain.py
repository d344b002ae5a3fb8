"""bf16 execution of a run of consecutive Dense layers (an MLP trunk).

A `Sequential` hands every maximal run of `Dense` layers to this module when
`config.compute_dtype() == "bf16"`.  Knowing the whole run lets each GEMM's
epilogue produce exactly what the next kernels consume, so no activation makes a
separate cast / transpose / activation-derivative pass through HBM:

  forward   layer l:  x_bf [M,K] . Wt_l  -> (+bias, act) -> y_bf [M,N] (next layer's
            operand, act' input and dW operand); the last layer also writes the
            fp32 chain output.
  backward  layer l:  dW_l, db_l from (x_bf_l, dz_bf_l) — both row-major, transposed
            on the fly inside the kernel;  then
            dz_{l-1} = (dz_l . W_l^T) ⊙ act'_{l-1}(y_{l-1}) written as bf16 by the
            dX kernel's epilogue.

Math per layer is `nnx_ppo/networks/feedforward.py:42-51` and its derivative;
products are bf16 x bf16 with fp32 accumulation, master weights stay fp32.
"""
from __future__ import annotations

import os
import weakref

import torch

from .. import ops
from .types import param_epoch


def _stale(layer) -> bool:
    w = layer.kernel.data
    return (getattr(layer, "_shadow_epoch", None) != param_epoch()
            or getattr(layer, "_w_bf", None) is None or layer._w_bf.device != w.device)


# layers that own bf16 shadows (weak: a dropped network drops its entries)
_SHADOWED: "weakref.WeakSet" = weakref.WeakSet()


def shadows_in(arena: torch.Tensor) -> list:
    """(layer, flat offset) of every shadowed layer whose fp32 kernel is a view of
    `arena` — the optimiser updates those images in its own launch (`mi_adam_step_f32`)."""
    lo = arena.data_ptr()
    hi = lo + arena.numel() * arena.element_size()
    out = []
    for l in list(_SHADOWED):
        w = l.kernel.data
        if getattr(l, "_w_bf", None) is None or l._w_bf.device != w.device:
            continue
        p = w.data_ptr()
        if lo <= p < hi and w.is_contiguous():
            out.append((l, (p - lo) // arena.element_size()))
    out.sort(key=lambda t: t[1])
    return out


def mark_fresh(layers) -> None:
    for l in layers:
        l._shadow_epoch = param_epoch()


def refresh(layers) -> None:
    """Refresh the bf16 shadows of every stale layer of a chain in one launch."""
    stale = [l for l in layers if _stale(l)]
    if not stale:
        return
    for l in stale:
        w = l.kernel.data
        K, N = w.shape
        _SHADOWED.add(l)
        if getattr(l, "_w_bf", None) is None or l._w_bf.device != w.device:
            l._w_bf = torch.zeros(K, ops.pad8(N), dtype=torch.bfloat16, device=w.device)
            l._wt_bf = torch.zeros(N, ops.pad8(K), dtype=torch.bfloat16, device=w.device)
            nf, nb = ops.frag_sizes(K, N)
            l._ff = torch.zeros(nf, dtype=torch.bfloat16, device=w.device)
            l._fb = torch.zeros(nb, dtype=torch.bfloat16, device=w.device)
    ops.weights_to_bf16_multi([l.kernel.data for l in stale], [l._w_bf for l in stale],
                              [l._wt_bf for l in stale], [l._ff for l in stale],
                              [l._fb for l in stale])
    for l in stale:
        l._shadow_epoch = param_epoch()


def _shadows(layer):
    """bf16 shadows (W [K, pad8 N], W^T [N, pad8 K]) of a Dense layer's fp32
    master kernel, refreshed when the parameters have changed."""
    if _stale(layer):
        refresh([layer])
    return layer._w_bf, layer._wt_bf


def _bias(layer):
    return layer.bias.data if layer.bias is not None else None


FUSED_MAX_LAYERS = 8
FUSED_MAX_WIDTH = 512


# MIPPO_GEMM256_CHAIN=1: trunks at least GEMM256_MIN_WIDTH wide run LAYER BY LAYER at training
# sizes (M > WS_MIN_ROWS) unless the weights-stationary kernels take them.  The per-layer NT
# GEMM on 256 x 256 tiles with direct-to-LDS loads (csrc/gemm256_bf16.hip) moves a
# 61 440 x 512 x 512 layer at ~650 TF/s with its bias / relu / image epilogue (the 128-row
# kernel: 385; hipBLASLt's plain product: 706) — but a trunk is not only its square layers:
# layer by layer, the thin first and last layers (17 -> 512, 512 -> 1) each make a full pass
# over a 63 MB image that the whole-trunk walk never writes twice, and BASELINE config 3 ran
# 18.4 M env-steps/s this way against 21.8 M on the whole-trunk kernels (measured, round 3).
# So the default stays the whole-trunk walk; trunks wider than FUSED_MAX_WIDTH, which always
# ran per layer, get the new kernel through the same entry points.
GEMM256_CHAIN = os.environ.get("MIPPO_GEMM256_CHAIN", "0") == "1"
GEMM256_MIN_WIDTH = 256


def _fusable(layers, M: int, need_input_grad: bool = False) -> bool:
    """Whole-trunk kernels (csrc/mlp_bf16.hip) take up to 8 layers of width <= 512.  Up to
    8192 rows (rollout / evaluation sizes) and for narrow trunks they beat the per-layer
    GEMMs (a layer is one or two k-tiles of work: launch and round-trip latency, not MFMA
    rate); wide trunks at training sizes go per layer (see GEMM256_CHAIN)."""
    width = max(max(l.in_features, l.out_features) for l in layers)
    if len(layers) > FUSED_MAX_LAYERS or width > FUSED_MAX_WIDTH:
        return False
    if (GEMM256_CHAIN and M > WS_MIN_ROWS and width >= GEMM256_MIN_WIDTH
            and not _ws(layers, M, need_input_grad)):
        return False
    return True


# Training-size chains in the shape class of the weights-stationary kernels (csrc/trunk_ws.hip:
# K0 <= 32 -> H -> (H -> H) x NH -> N <= 16, relu, linear head) run on them: one persistent
# 8-wave workgroup per CU with the trunk in registers, bit-identical to the whole-trunk tile
# kernels (tests/test_trunk_ws_gpu.py), 21 -> 14 us for the 5-256-256-1 critic at M = 30 720.
# Below WS_MIN_ROWS the tile kernels (one workgroup per 64 rows) fill the chip better.
WS_CHAIN = os.environ.get("MIPPO_WS_CHAIN", "1") != "0"
WS_MIN_ROWS = 8192


def _ws(layers, M: int, need_input_grad: bool) -> bool:
    if not WS_CHAIN or need_input_grad or M <= WS_MIN_ROWS or len(layers) < 2:
        return False
    dims = [layers[0].in_features] + [l.out_features for l in layers]
    return ops.mlp_ws_supported(dims, [l.act_code for l in layers])


def _chain_args(layers):
    refresh(layers)
    wts = [l._ff for l in layers]  # forward fragment-major images
    biases = [_bias(l) for l in layers]
    dims = [layers[0].in_features] + [l.out_features for l in layers]
    acts = [l.act_code for l in layers]
    return wts, biases, dims, acts


def forward_infer(layers, x2: torch.Tensor) -> torch.Tensor:
    """fp32 [M, K0] -> fp32 [M, N_last]; no activations are kept."""
    if _fusable(layers, x2.shape[0]):
        out, _ = ops.mlp_fwd_bf16(x2, *_chain_args(layers), train=False)
        return out
    refresh(layers)
    x_bf = ops.cast_pad_bf16(x2)
    y = None
    for i, layer in enumerate(layers):
        last = i == len(layers) - 1
        _, wt = _shadows(layer)
        y, y_bf, _ = ops.dense_fwd_bf16(x_bf, wt, _bias(layer), layer.in_features,
                                        layer.out_features, layer.act_code, want_f32=last,
                                        want_bf=not last)
        x_bf = y_bf
    return y


def forward_train(layers, x2: torch.Tensor, need_input_grad: bool, want_out: bool = True):
    """Returns (ctx, fp32 output [M, N_last]).  ctx keeps, per layer, the bf16 input
    (dW operand), the tensor its act' is evaluated on, and the bf16 W shadow.
    `want_out=False`: the caller reads the last layer's bf16 image (ctx) only; the fp32 output
    may come back as None."""
    M = x2.shape[0]
    if _fusable(layers, M, need_input_grad):
        if _ws(layers, M, need_input_grad):
            y, sv = ops.mlp_ws_fwd_bf16(x2, *_chain_args(layers), train=True)
        else:
            y, sv = ops.mlp_fwd_bf16(x2, *_chain_args(layers), train=True,
                                     want_out=want_out or layers[-1].act_code == ops.ACT_NONE)
        saved = [(xb, aux, _shadows(l)[0]) for (xb, aux), l in zip(sv, layers)]
        return (saved, M, need_input_grad), y
    refresh(layers)
    x_bf = ops.cast_pad_bf16(x2)
    saved = []
    y = None
    for i, layer in enumerate(layers):
        last = i == len(layers) - 1
        act = layer.act_code
        swish = act == ops.ACT_SWISH
        w_bf, wt = _shadows(layer)
        y, y_bf, pre = ops.dense_fwd_bf16(
            x_bf, wt, _bias(layer), layer.in_features, layer.out_features, act,
            want_f32=last, want_bf=(not last) or act != ops.ACT_NONE, want_preact=swish)
        saved.append((x_bf, pre if swish else y_bf, w_bf))
        x_bf = y_bf
    return (saved, M, need_input_grad), y


def backward(layers, ctx, g_out2: torch.Tensor):
    """g_out2: fp32 [M, N_last] gradient of the chain output.  Accumulates weight /
    bias gradients; returns the fp32 input gradient [M, K0] or None.

    The dX chain runs first (it is sequential by nature: one fused launch when the
    trunk fits the whole-trunk kernel) and keeps every dz; the dW / db of ALL
    layers then go out as one grouped launch per tile class."""
    saved, M_fwd, need_input_grad = ctx
    M = g_out2.shape[0]
    if M != M_fwd:
        # the backward may cover a PREFIX of the rows the forward saw (bootstrap rows that rode
        # along in a value chain's launch carry no gradient): row-prefix views of the images
        assert M < M_fwd, "dense_chain.backward: more gradient rows than forward rows"
        saved = [tuple(None if t is None else t[:M] for t in sv[:2]) + tuple(sv[2:])
                 for sv in saved]
    L = len(layers)
    last = layers[-1]
    grads = [(l.kernel.grad, l.bias.grad if l.bias is not None else None) for l in layers]
    if _fusable(layers, M_fwd, need_input_grad) and (L > 1 or need_input_grad):
        dims = [layers[0].in_features] + [l.out_features for l in layers]
        refresh(layers)
        if _ws(layers, M_fwd, need_input_grad):
            dz = ops.mlp_ws_bwd_dx_bf16(g_out2, [l._fb for l in layers], dims,
                                        [l.act_code for l in layers], [sv[1] for sv in saved])
            ops.dense_bwd_dw_grouped_bf16(
                [(saved[i][0], dz[i], grads[i][0], grads[i][1]) for i in range(L - 1, -1, -1)],
                accumulate=True)
            return None
        dz, g_in = ops.mlp_bwd_dx_bf16(
            g_out2, saved[-1][1] if last.act_code != ops.ACT_NONE else None, last.act_code,
            [l._fb for l in layers], dims, [l.act_code for l in layers],
            [sv[1] for sv in saved], need_input_grad)
        ops.dense_bwd_dw_grouped_bf16(
            [(saved[i][0], dz[i], grads[i][0], grads[i][1]) for i in range(L - 1, -1, -1)],
            accumulate=True)
        return g_in
    dz_bf = ops.cast_pad_bf16(g_out2, aux=saved[-1][1] if last.act_code != ops.ACT_NONE else None,
                              act=last.act_code)
    g_in = None
    problems = []
    for i in range(L - 1, -1, -1):
        layer = layers[i]
        x_bf, _, w_bf = saved[i]
        problems.append((x_bf, dz_bf, grads[i][0], grads[i][1]))
        if i == 0:
            if need_input_grad:
                g_in, _ = ops.dense_bwd_dx_bf16(dz_bf, w_bf, None, ops.ACT_NONE,
                                                layer.in_features, layer.out_features,
                                                want_f32=True, want_bf=False)
            break
        prev = layers[i - 1]
        _, dz_bf = ops.dense_bwd_dx_bf16(dz_bf, w_bf, saved[i - 1][1], prev.act_code,
                                         layer.in_features, layer.out_features,
                                         want_f32=False, want_bf=True)
    ops.dense_bwd_dw_grouped_bf16(problems, accumulate=True)
    return g_in
