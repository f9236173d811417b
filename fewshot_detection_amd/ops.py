"""Thin tensor-level wrappers over the C-ABI.  Every function launches on the current torch HIP
stream and never synchronises.  Tensors are raw storage here: layouts are documented per op."""
import os

import torch

from ._lib import check, lib


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def upload_words(src_pinned, dst, words):
    """dst (device) <- `words` 4-byte words of src_pinned (page-locked host tensor), by a kernel on the current stream."""
    if dst is None or not dst.is_cuda:
        raise RuntimeError("upload_words: the destination must be a HIP device tensor")
    if src_pinned.is_cuda or not src_pinned.is_pinned():
        raise RuntimeError("upload_words: the source must be a page-locked host tensor")
    check(lib().fsd_upload_words(src_pinned.data_ptr(), dst.data_ptr(), int(words), _stream()), "fsd_upload_words")


def require_device(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("fewshot_detection_amd ops need HIP device tensors (got a %s tensor); "
                               "there is no CPU fallback" % t.device)


class View(object):
    """NHWC activation: `t` is a 2-D (pixels, ld) buffer, channels [c0, c0+C) of each pixel.  The buffer is float32
    (fp32 mode) or bfloat16 (bf16 storage mode); `ld` and `c0` count ELEMENTS.
    `lazy` = (scale, shift, slope) marks a DEFERRED activation (fp32): the buffer holds the raw output y of a conv and the
    view stands for leaky(y * scale + shift); consumers that can form it on load do (conv2d, conv3x3_wino tile 4, the 1x1
    weight gradient), everybody else calls materialise() first."""
    __slots__ = ("t", "B", "H", "W", "C", "c0", "lazy")

    def __init__(self, t, B, H, W, C, c0=0, lazy=None):
        self.t, self.B, self.H, self.W, self.C, self.c0, self.lazy = t, B, H, W, C, c0, lazy

    @property
    def ld(self):
        return self.t.shape[1]

    @property
    def bf16(self):
        return self.t.dtype == torch.bfloat16

    @property
    def ptr(self):
        return self.t.data_ptr() + self.t.element_size() * self.c0

    @property
    def pixels(self):
        return self.B * self.H * self.W

    def dense(self):
        return self.t[:, self.c0:self.c0 + self.C]


def materialise(v, out=None):
    """The activation a deferred view stands for, as a plain view (one BatchNorm + leaky pass); plain views pass through."""
    if v.lazy is None:
        return v
    scale, shift, slope = v.lazy
    return bn_act_pool(View(v.t, v.B, v.H, v.W, v.C, v.c0), scale, shift, slope, 0, out=out)


def _plain(*views):
    """Guard of the consumers that read a view's buffer as the activation itself: a deferred view holds the producing
    convolution's RAW output (see View.lazy) -- materialise() it first."""
    for v in views:
        if v is not None and v.lazy is not None:
            raise ValueError("this operation reads the activation itself; the view is deferred (ops.materialise() it first)")


def lazy_ok_direct(v, ksize):
    """Can conv2d / the 1x1 weight gradient form this deferred activation on load?  (whole 32-channel chunks per tap)"""
    return v.lazy is not None and not v.bf16 and v.C % 32 == 0 and ksize in (1, 3)


def new_view(B, H, W, C, device, ld=None, dtype=torch.float32):
    return View(torch.empty((B * H * W, ld or C), dtype=dtype, device=device), B, H, W, C)


def like_view(v, C=None, ld=None, H=None, W=None):
    """A fresh view with the storage type of `v` (fp32 or bf16 mode follows the activations)."""
    return new_view(v.B, v.H if H is None else H, v.W if W is None else W, v.C if C is None else C, v.t.device, ld=ld,
                    dtype=v.t.dtype)


def cast_view(v, dtype, out=None):
    """A dense copy of `v` in another storage type (float32 <-> bfloat16, round-to-nearest-even).  Plumbing for the few
    convolutions of non-standard cfgs whose channel counts the bf16 kernels do not take; the shipped cfgs never call it."""
    _plain(v)
    src = v.t[:, v.c0:v.c0 + v.C]
    if out is None:
        return View(src.to(dtype).contiguous(), v.B, v.H, v.W, v.C)
    out.t[:, out.c0:out.c0 + out.C].copy_(src)
    return out


def fill(t, value):
    check(lib().fsd_fill(t.data_ptr(), float(value), t.numel(), _stream()), "fsd_fill")
    return t


def nchw_to_nhwc(x, pad_to=4, out=None, dtype=torch.float32):
    """(B,C,H,W) contiguous float32 -> View with channels padded (zeros) to a multiple of `pad_to`; `dtype` = storage
    type of the result (torch.bfloat16 in bf16 mode)."""
    require_device(x)
    if x.dtype != torch.float32:
        raise ValueError("nchw_to_nhwc needs a float32 tensor (got %s)" % x.dtype)
    x = x.contiguous()
    B, Cc, H, W = x.shape
    Cp = (Cc + pad_to - 1) // pad_to * pad_to
    if out is None and Cp == 4 and dtype == torch.float32:            # network inputs: one pass, zero padding included
        out = new_view(B, H, W, 4, x.device)
        check(lib().fsd_nchw_to_nhwc4(x.data_ptr(), out.ptr, B, Cc, H * W, _stream()), "fsd_nchw_to_nhwc4")
        return out
    if out is None:
        out = new_view(B, H, W, Cp, x.device, dtype=dtype)
        if Cp != Cc:
            out.t.zero_()
    if out.bf16:
        check(lib().fsd_transpose_batched_h(x.data_ptr(), 0, Cc * H * W, H * W, out.ptr, 1, H * W * out.ld, out.ld,
                                            B, Cc, H * W, _stream()), "fsd_transpose_batched_h")
    else:
        check(lib().fsd_transpose_batched(x.data_ptr(), Cc * H * W, H * W, out.ptr, H * W * out.ld, out.ld,
                                          B, Cc, H * W, _stream()), "fsd_transpose_batched")
    return out


def write_channels(x, view, c_off):
    """Scatter an NCHW tensor into channels [c_off, c_off+C) of an existing NHWC view."""
    B, Cc, H, W = x.shape
    if x.dtype != torch.float32:
        raise ValueError("write_channels needs a float32 tensor (got %s)" % x.dtype)
    if (B, H, W) != (view.B, view.H, view.W) or c_off < 0 or c_off + Cc > view.ld - view.c0:
        raise ValueError("write_channels: a %s tensor at channel %d does not fit the (%d, %d, %d, %d) view"
                         % (tuple(x.shape), c_off, view.B, view.H, view.W, view.C))
    x = x.contiguous()
    check(lib().fsd_transpose_batched(x.data_ptr(), Cc * H * W, H * W, view.ptr + 4 * c_off, H * W * view.ld,
                                      view.ld, B, Cc, H * W, _stream()), "fsd_transpose_batched")


def nhwc_to_nchw(v):
    _plain(v)
    out = torch.empty((v.B, v.C, v.H, v.W), dtype=torch.float32, device=v.t.device)
    hw = v.H * v.W
    if v.bf16:
        check(lib().fsd_transpose_batched_h(v.ptr, 1, hw * v.ld, v.ld, out.data_ptr(), 0, v.C * hw, hw, v.B, hw, v.C,
                                            _stream()), "fsd_transpose_batched_h")
    else:
        check(lib().fsd_transpose_batched(v.ptr, hw * v.ld, v.ld, out.data_ptr(), v.C * hw, hw, v.B, hw, v.C,
                                          _stream()), "fsd_transpose_batched")
    return out


def _reuse(out, numel, dtype, device):
    """`out` if it is a buffer of exactly this size / type / device (a packed copy from an earlier step, rewritten in place:
    the weights' readers are stream-ordered behind the optimizer step that precedes a re-pack), else a fresh one."""
    if out is not None and out.numel() == numel and out.dtype == dtype and out.device == device:
        return out
    return torch.empty(numel, dtype=dtype, device=device)


def pack_weight(w, mode=0, dtype="f32", out=None):
    """(Cout,Cin,k,k) -> packed K-major operand of fsd_conv2d_fwd[_bf16] (mode 0) / its data gradient (mode 1).
    out: the packed buffer of an earlier call for the same weight (rewritten in place: no allocator traffic per step)."""
    require_device(w)
    cout, cin, k, _ = w.shape
    rows, red = (cout, cin) if mode == 0 else (cin, cout)
    if dtype == "bf16":
        out = _reuse(out, lib().fsd_packed_weight_elems_bf16(rows, red, k), torch.bfloat16, w.device)
        check(lib().fsd_pack_conv_weight_bf16(w.contiguous().data_ptr(), out.data_ptr(), cout, cin, k, mode, _stream()),
              "fsd_pack_conv_weight_bf16")
        return out
    out = _reuse(out, lib().fsd_packed_weight_elems(rows, red, k), torch.float32, w.device)
    check(lib().fsd_pack_conv_weight(w.contiguous().data_ptr(), out.data_ptr(), cout, cin, k, mode, _stream()),
          "fsd_pack_conv_weight")
    return out


def pack_weight_bf16_pair(w, out=None):
    """Forward (mode 0) and data-gradient (mode 1) bf16 operands of one conv weight in a single pass.  out: a pair of
    buffers from an earlier call (re-used: their zero padding never changes)."""
    require_device(w)
    cout, cin, k, _ = w.shape
    if out is None:
        out = (torch.zeros(lib().fsd_packed_weight_elems_bf16(cout, cin, k), dtype=torch.bfloat16, device=w.device),
               torch.zeros(lib().fsd_packed_weight_elems_bf16(cin, cout, k), dtype=torch.bfloat16, device=w.device))
    check(lib().fsd_pack_conv_weight_bf16_pair(w.contiguous().data_ptr(), out[0].data_ptr(), out[1].data_ptr(), cout, cin, k,
                                               _stream()), "fsd_pack_conv_weight_bf16_pair")
    return out


def pack_weight_wino(w, mode=0, tile=2, out=None):
    """(Cout,Cin,3,3) -> the (tile+2)^2 transformed (G g G^T) matrices in the GEMM kernel's packed layout.
    out: the buffer of an earlier call for the same weight (rewritten in place)."""
    require_device(w)
    cout, cin, k, _ = w.shape
    assert k == 3
    rows, red = (cout, cin) if mode == 0 else (cin, cout)
    out = _reuse(out, lib().fsd_wino_packed_weight_elems(rows, red, tile), torch.float32, w.device)
    check(lib().fsd_wino_pack_weight(w.contiguous().data_ptr(), out.data_ptr(), cout, cin, mode, tile, _stream()),
          "fsd_wino_pack_weight")
    return out


def conv3x3_wino(xv, u_packed, cout, bias=None, out=None, bn_partial=False, keep_v=None, tile=2, v_in=None, slope=1.0):
    """Winograd F(tile x tile, 3x3) convolution of an NHWC view; same results/contract as conv2d(ksize=3).
    slope != 1: y = leaky(conv + bias) (inference form, no statistics).
    keep_v: a list; the transformed input is appended to it (kept for the weight gradient).
    v_in: an already transformed input (wino_grad_transforms); xv then only supplies the geometry."""
    L = lib()
    dev = xv.t.device
    y = out if out is not None else new_view(xv.B, xv.H, xv.W, cout, dev)
    partial = None
    if bn_partial:
        partial = torch.empty((L.fsd_wino_partial_rows(xv.B, xv.H, xv.W, tile), cout, 2), dtype=torch.float32,
                              device=dev)
    ws_bytes = L.fsd_wino_workspace_bytes(xv.B, xv.H, xv.W, xv.C, cout, tile)
    ws = torch.empty((ws_bytes + 3) // 4, dtype=torch.float32, device=dev)
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    v = None
    if keep_v is not None:
        v = torch.empty(L.fsd_wino_v_elems(xv.B, xv.H, xv.W, xv.C, tile), dtype=torch.float32, device=dev)
        keep_v.append(v)
    if xv.lazy is not None and v_in is None:
        if tile != 4:
            raise ValueError("a deferred activation needs the F(4x4) input transform (materialise() it first)")
        sc, sh, sl = xv.lazy
        check(L.fsd_wino_conv3x3_fwd_ex(xv.ptr, xv.ld, u_packed.data_ptr(), _ptr(bias), y.ptr, y.ld, _ptr(partial),
                                        ws.data_ptr(), ws_bytes, _ptr(v), 0, xv.B, xv.H, xv.W, xv.C, cout, tile,
                                        float(slope), sc.data_ptr(), sh.data_ptr(), float(sl), _stream()),
              "fsd_wino_conv3x3_fwd_ex")
    else:
        check(L.fsd_wino_conv3x3_fwd_act(xv.ptr, xv.ld, u_packed.data_ptr(), _ptr(bias), y.ptr, y.ld, _ptr(partial),
                                         ws.data_ptr(), ws_bytes, _ptr(v), _ptr(v_in), xv.B, xv.H, xv.W, xv.C, cout, tile,
                                         float(slope), _stream()), "fsd_wino_conv3x3_fwd")
    if PROFILE is not None:
        e1.record()
        tiles = xv.B * ((xv.H + tile - 1) // tile) * ((xv.W + tile - 1) // tile)
        PROFILE.append((e0, e1, 2.0 * 9 * xv.C * cout * xv.pixels, 2.0 * (tile + 2) ** 2 * xv.C * cout * tiles,
                        4.0 * (xv.pixels * (xv.C + cout) + 9 * xv.C * cout)))
    return y, partial


def wino_eligible(cin, cout, ksize):
    """F(2x2,3x3) rule.  Measured on MI355X: the transforms cost ~16x(Cin+Cout) floats of HBM traffic per tile, which
    only pays once both channel counts reach 128 (the 104x104 layers with 64 channels are faster on the direct kernel)."""
    return WINOGRAD and ksize == 3 and cin % 32 == 0 and cout % 4 == 0 and min(cin, cout) >= 128


def wino_tile(cin, cout, ksize, H, W):
    """Which form a 3x3 fp32 convolution takes: 0 = direct implicit GEMM, 2 = F(2x2,3x3), 4 = F(4x4,3x3).
    F(4x4) needs 36 multiplications per 4x4 tile against 16 per 2x2 tile, so it wins when the feature map does not
    waste too much of its 4x4 tiles (13x13 -> 16x16: 0.73 of the F(2x2) work; 26x26: 0.65; 52x52: 0.56)."""
    if not (WINOGRAD and ksize == 3 and cin % 32 == 0 and cout % 4 == 0):
        return 0
    lo = min(cin, cout)
    # (round 5: the 64 <-> 128 layers at 104x104 on a halo-staged DIRECT kernel instead -- their unfused F(4x4) pipeline moves
    # 5.6x the layer's bytes, profiles/r05_traffic_ledger.csv -- measured: 0.49 ms forward + 0.43 ms data gradient per layer
    # and a step 0.7 ms slower (the six-term direct arithmetic costs what the transforms' bytes cost);
    # tools/experiments_r05/halo_direct_64_128_f32.patch)
    m2 = 16 * ((H + 1) // 2) * ((W + 1) // 2)
    m4 = 36 * ((H + 3) // 4) * ((W + 3) // 4)
    if WINOGRAD4 and lo >= WINO4_MIN_CH and m4 <= 0.85 * m2:
        return 4
    return 2 if lo >= 128 else 0


def wino_tile_inference(cin, cout, ksize, H, W, batch):
    """The inference form's choice (BatchNorm folded, nothing kept for a backward pass): wino_tile's, except that a position
    GEMM of at most 64 rows is bound by READING its transformed weights -- 36 x Cin x Cout floats for F(4x4) against 16 for
    F(2x2): 151 MB against 67 MB on a 1024 -> 1024 layer, 30 us against 13 at the transforms' 5 TB/s, which is more than
    the layer's arithmetic at valid_ensemble.py's two images per batch.  OPT-IN (SMALL_BATCH_F2 = True): measured on MI355X the
    forward of two images went from 0.83 to 0.88 ms with it -- the F(4x4) launches are short of bytes in flight, not of
    bandwidth (fsd_conv::batched_ksplit is the fix that worked), and the F(2x2) transforms are the older, slower kernels."""
    tile = wino_tile(cin, cout, ksize, H, W)
    if (tile == 4 and SMALL_BATCH_F2 and batch * ((H + 3) // 4) * ((W + 3) // 4) <= 64 and cin * cout >= 512 * 512):
        return 2
    return tile


SMALL_BATCH_F2 = False
WINOGRAD = True     # Winograd for eligible fp32 3x3 layers (forward, data gradient, weight gradient)
WINOGRAD4 = True    # allow F(4x4,3x3) where it needs fewer multiplications than F(2x2,3x3)
# One-pass BN backward + both gradient transforms (fsd_wino_grad_transforms).  Bit-identical to the separate kernels but
# MEASURED SLOWER on MI355X (3.59 vs 3.04 ms per step: 72 loads + 72 stores per thread over two overlapping 6x6 patches
# run at 4.5 TB/s at L2 level against 7.8 TB/s for the single-tensor transforms), so it is off by default.
FUSE_WINO_GRAD = False
WINO4_MIN_CH = 64   # F(4x4): minimum of (Cin, Cout) (measured: pays from 64 channels at 104x104, not at 32 / 208x208)
PROFILE = None      # bench.py sets this to a list, one entry per conv launch (forward / data gradient; a Winograd launch =
                    # transform + GEMM + transform): (start_event, end_event, algorithmic_flops, executed_mfma_flops,
                    # algorithmic_hbm_bytes = input + output activation + weights, each once).
                    # Per-KERNEL timing is the library's job (fsd_profile_enable / fsd_profile_collect).


def conv2d(xv, w_packed, cout, ksize, bias=None, out=None, bn_partial=False, nchw_out=False, cin_true=None, slope=1.0):
    """xv: View (C % 4 == 0).  Returns (y, partial): y is a View (or an NCHW tensor if nchw_out).
    slope != 1: y = leaky(conv + bias) (inference form: NHWC store, no statistics)."""
    dev = xv.t.device
    partial = None
    if xv.bf16:
        return _conv2d_h(xv, w_packed, cout, ksize, bias, out, bn_partial, nchw_out, slope)
    if nchw_out:
        y = torch.empty((xv.B, cout, xv.H, xv.W), dtype=torch.float32, device=dev)
        y_ptr, y_ld = y.data_ptr(), 0
    else:
        y = out if out is not None else new_view(xv.B, xv.H, xv.W, cout, dev)
        y_ptr, y_ld = y.ptr, y.ld
    if w_packed.dtype != torch.float32:
        raise ValueError("fp32 activations need fp32-packed weights (bf16 weights belong to bf16 activations)")
    if bn_partial:
        tiles = lib().fsd_conv_row_tiles(xv.B, xv.H, xv.W, cout, xv.C, ksize)
        partial = torch.empty((tiles, cout, 2), dtype=torch.float32, device=dev)
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    if xv.lazy is not None:
        if not lazy_ok_direct(xv, ksize):
            raise ValueError("this convolution cannot form a deferred activation on load (materialise() it first)")
        sc, sh, sl = xv.lazy
        check(lib().fsd_conv2d_fwd_ex(xv.ptr, xv.ld, w_packed.data_ptr(), _ptr(bias), y_ptr, y_ld, _ptr(partial),
                                      xv.B, xv.H, xv.W, xv.C, cout, ksize, 1 if nchw_out else 0, float(slope),
                                      sc.data_ptr(), sh.data_ptr(), float(sl), _stream()), "fsd_conv2d_fwd_ex")
    else:
        check(lib().fsd_conv2d_fwd_act(xv.ptr, xv.ld, w_packed.data_ptr(), _ptr(bias), y_ptr, y_ld, _ptr(partial),
                                       xv.B, xv.H, xv.W, xv.C, cout, ksize, 1 if nchw_out else 0, float(slope), _stream()),
              "fsd_conv2d_fwd")
    if PROFILE is not None:
        e1.record()
        PROFILE.append((e0, e1, 2.0 * ksize * ksize * (cin_true or xv.C) * cout * xv.pixels,
                        2.0 * ksize * ksize * xv.C * cout * xv.pixels,
                        4.0 * (xv.pixels * (xv.C + cout) + ksize * ksize * xv.C * cout)))
    return y, partial


def _conv2d_h(xv, w_packed, cout, ksize, bias, out, bn_partial, nchw_out, slope=1.0):
    """bf16 storage mode: bf16 NHWC activations x packed bf16 weights -> bf16 NHWC (or float NCHW for the head)."""
    _plain(xv)
    L = lib()
    dev = xv.t.device
    if w_packed.dtype != torch.bfloat16:
        raise ValueError("bf16 activations need bf16-packed weights")
    partial = None
    if nchw_out:
        y = torch.empty((xv.B, cout, xv.H, xv.W), dtype=torch.float32, device=dev)
        y_ptr, y_ld = y.data_ptr(), 0
    else:
        y = out if out is not None else new_view(xv.B, xv.H, xv.W, cout, dev, dtype=torch.bfloat16)
        if not y.bf16:
            raise ValueError("bf16 convolution needs a bf16 output view")
        y_ptr, y_ld = y.ptr, y.ld
    if bn_partial:
        # (the plan for THESE operands: a view with an odd pixel stride takes the GEMM kernel's row tiles)
        partial = torch.empty((L.fsd_conv2d_h_partial_rows_at(xv.B, xv.H, xv.W, xv.C, cout, ksize, xv.ptr, xv.ld, y_ptr, y_ld),
                               cout, 2), dtype=torch.float32, device=dev)
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    check(L.fsd_conv2d_fwd_act_h(xv.ptr, xv.ld, w_packed.data_ptr(), _ptr(bias), y_ptr, y_ld, _ptr(partial), xv.B, xv.H, xv.W,
                                 xv.C, cout, ksize, 1 if nchw_out else 0, float(slope), _stream()), "fsd_conv2d_fwd_h")
    if PROFILE is not None:
        e1.record()
        fl = 2.0 * ksize * ksize * xv.C * cout * xv.pixels
        PROFILE.append((e0, e1, fl, fl, 2.0 * xv.pixels * xv.C + (4.0 if nchw_out else 2.0) * xv.pixels * cout
                        + 2.0 * ksize * ksize * xv.C * cout))
    return y, partial


def conv3x3_c4(xv, w, cout, bias=None, out=None, bn_partial=False, out_dtype=torch.float32):
    """First-layer 3x3 convolution of an NHWC4 view straight from the OIHW weights (HBM-bound direct-operand kernel).
    out_dtype = torch.bfloat16 stores the raw output in bf16 (bf16 mode); the arithmetic is fp32 either way."""
    _plain(xv)
    L = lib()
    dev = xv.t.device
    y = out if out is not None else new_view(xv.B, xv.H, xv.W, cout, dev, dtype=out_dtype)
    partial = None
    if bn_partial:
        partial = torch.empty((L.fsd_conv3x3_c4_partial_rows(xv.B, xv.H, xv.W), cout, 2), dtype=torch.float32, device=dev)
    cin = w.shape[1]
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    fn = L.fsd_conv3x3_c4_fwd_h if y.bf16 else L.fsd_conv3x3_c4_fwd
    check(fn(xv.ptr, xv.ld, w.detach().contiguous().data_ptr(), _ptr(bias), y.ptr, y.ld, _ptr(partial),
             xv.B, xv.H, xv.W, cin, cout, _stream()), "fsd_conv3x3_c4_fwd")
    if PROFILE is not None:
        e1.record()
        PROFILE.append((e0, e1, 2.0 * 9 * cin * cout * xv.pixels, 2.0 * 9 * 4 * cout * xv.pixels,
                        xv.pixels * (16.0 + y.t.element_size() * cout) + 4.0 * 9 * cin * cout))
    return y, partial


def bn_finalize(partial, count, bn, training):
    """-> (scale, shift, save_mean, save_invstd) for nn.BatchNorm2d-like `bn` (updates running stats)."""
    Cc = bn.num_features
    dev = bn.weight.device
    buf = torch.empty((4, Cc), dtype=torch.float32, device=dev)
    ws = None
    tiles = 0
    if training:
        ws = torch.empty(lib().fsd_bn_finalize_workspace_bytes(Cc) // 8, dtype=torch.float64, device=dev)
        tiles = partial.shape[0]
    momentum = 0.1 if bn.momentum is None else bn.momentum
    check(lib().fsd_bn_finalize(_ptr(partial), tiles, count, Cc, bn.weight.data_ptr(), bn.bias.data_ptr(),
                                bn.running_mean.data_ptr(), bn.running_var.data_ptr(), momentum, bn.eps,
                                1 if training else 0, buf[0].data_ptr(), buf[1].data_ptr(), buf[2].data_ptr(),
                                buf[3].data_ptr(), _ptr(ws), _stream()), "fsd_bn_finalize")
    if training:
        from .engine import bump_stats_epoch
        bump_stats_epoch()               # running statistics changed behind torch's version counters
        # nn.BatchNorm2d counts its training batches in a device buffer (torch >= 0.4; only read when momentum is None):
        # counted on the host here and written back when someone looks (flush_bn_counters, called by state_dict())
        bn._fsd_pending_batches = getattr(bn, "_fsd_pending_batches", 0) + 1
    return buf[0], buf[1], buf[2], buf[3]


def _bn_counter_flush_hook(m, prefix, keep_vars):
    n = getattr(m, "_fsd_pending_batches", 0)
    if n and getattr(m, "num_batches_tracked", None) is not None:
        m.num_batches_tracked += n
    m._fsd_pending_batches = 0


def _bn_counter_drop_hook(m, incompatible):
    m._fsd_pending_batches = 0


def install_bn_counter_hooks(module):
    """Every BatchNorm of `module` flushes its host-side batch count whenever ITS state is gathered (state_dict() of the
    module itself or of any parent, torch.save(model.models.state_dict()) included), and drops the pending count when a state
    is loaded into it (load_state_dict: the loaded counter is the truth, batches run before the load must not be added on
    top; load_weights does not touch the counter -- a .weights file has none -- so the pending count stays, as in the reference).
    The hooks are module-level functions: a model that carries them still pickles (torch.save(model), spawned workers)."""
    for m in module.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm) and not getattr(m, "_fsd_counter_hooks", False):
            m.register_state_dict_pre_hook(_bn_counter_flush_hook)
            m.register_load_state_dict_post_hook(_bn_counter_drop_hook)
            m._fsd_counter_hooks = True


def flush_bn_counters(module):
    """Bring every BatchNorm's `num_batches_tracked` up to date with the training batches the HIP path ran."""
    for m in module.modules():
        n = getattr(m, "_fsd_pending_batches", 0)
        if n and getattr(m, "num_batches_tracked", None) is not None:
            m.num_batches_tracked += n
            m._fsd_pending_batches = 0


def bn_act_pool(yv, scale, shift, slope, pool, out=None):
    _plain(yv)
    OH, OW = (yv.H // 2, yv.W // 2) if pool == 1 else (yv.H, yv.W)
    z = out if out is not None else like_view(yv, H=OH, W=OW)
    if z.bf16 != yv.bf16:
        raise ValueError("bn_act_pool: input and output views must share the storage type")
    fn = lib().fsd_bn_act_pool_fwd_h if yv.bf16 else lib().fsd_bn_act_pool_fwd
    check(fn(yv.ptr, yv.ld, _ptr(scale), _ptr(shift), slope, pool, z.ptr, z.ld,
             yv.B, yv.H, yv.W, yv.C, _stream()), "fsd_bn_act_pool_fwd")
    return z


def reorg(xv, stride, out=None):
    _plain(xv)
    z = out if out is not None else like_view(xv, C=xv.C * stride * stride, H=xv.H // stride, W=xv.W // stride)
    fn = lib().fsd_reorg_fwd_h if xv.bf16 else lib().fsd_reorg_fwd
    check(fn(xv.ptr, xv.ld, z.ptr, z.ld, xv.B, xv.H, xv.W, xv.C, stride, _stream()), "fsd_reorg_fwd")
    return z


def global_maxpool(xv, want_argmax=False):
    _plain(xv)
    out = torch.empty((xv.B, xv.C), dtype=torch.float32, device=xv.t.device)
    arg = torch.empty((xv.B, xv.C), dtype=torch.int32, device=xv.t.device) if want_argmax else None
    fn = lib().fsd_global_maxpool_fwd_h if xv.bf16 else lib().fsd_global_maxpool_fwd
    check(fn(xv.ptr, xv.ld, out.data_ptr(), _ptr(arg), xv.B, xv.H, xv.W, xv.C, _stream()), "fsd_global_maxpool_fwd")
    return out, arg


def global_avgpool(xv):
    """(B,H,W,C) view -> (B,C) float: the mean over the map (pooling.GlobalAvgPool2d, [globalavg] / [avgpool] blocks)."""
    _plain(xv)
    out = torch.empty((xv.B, xv.C), dtype=torch.float32, device=xv.t.device)
    check(lib().fsd_global_avgpool_fwd(xv.ptr, 1 if xv.bf16 else 0, xv.ld, out.data_ptr(), xv.B, xv.H, xv.W, xv.C, _stream()),
          "fsd_global_avgpool_fwd")
    return out


def global_avgpool_bwd(dout, xv):
    dx = like_view(xv)
    check(lib().fsd_global_avgpool_bwd(dout.contiguous().data_ptr(), dx.ptr, 1 if dx.bf16 else 0, dx.ld, xv.B, xv.H, xv.W, xv.C,
                                       _stream()), "fsd_global_avgpool_bwd")
    return dx


def dynamic_conv(x, w):
    """Materialising reweighting on NCHW tensors: out[b*N+n,c,h,w] = x[b,c,h,w] * w[n,c]."""
    require_device(x, w)
    B, Cc, H, W = x.shape
    N = w.shape[0]
    out = torch.empty((B * N, Cc, H, W), dtype=torch.float32, device=x.device)
    check(lib().fsd_dynamic_conv_fwd(x.contiguous().data_ptr(), w.contiguous().data_ptr(), out.data_ptr(),
                                     B, N, Cc, H * W, _stream()), "fsd_dynamic_conv_fwd")
    return out


def fold_reweight_head(head_w, head_b, dyn, dtype="f32"):
    """-> (w_eff_packed, bias_eff, w_eff_f32) for the fused reweighting (x) 1x1 head GEMM.  The fp32 fold is
    always built (the backward un-folds it); in bf16 mode it is re-packed as the bf16 operand."""
    w_eff, b_eff = _fold_reweight_head_f32(head_w, head_b, dyn)
    if dtype == "bf16":
        rows, Cc = dyn.shape[0] * head_w.shape[0], head_w.shape[1]
        return pack_weight(w_eff[:rows * Cc].view(rows, Cc, 1, 1), 0, "bf16"), b_eff, w_eff
    return w_eff, b_eff, w_eff


def _fold_reweight_head_f32(head_w, head_b, dyn):
    O, Cc = head_w.shape[0], head_w.shape[1]
    N = dyn.shape[0]
    w_eff = torch.empty(lib().fsd_packed_weight_elems(N * O, Cc, 1), dtype=torch.float32, device=dyn.device)
    b_eff = torch.empty(N * O, dtype=torch.float32, device=dyn.device)
    check(lib().fsd_fold_reweight_head(head_w.contiguous().data_ptr(), _ptr(head_b), dyn.contiguous().data_ptr(),
                                       w_eff.data_ptr(), b_eff.data_ptr(), N, O, Cc, _stream()),
          "fsd_fold_reweight_head")
    return w_eff, b_eff


# ---- backward ------------------------------------------------------------------------------

GRAD_SINK = None    # dp.EpisodeTrainer: {id(param): 1-D view into its flat gradient buffer} while a step's backward runs
GRAD_SUNK = set()   # ids of the parameters whose gradient the kernels wrote straight into the sink
GRAD_HOOK = None    # called after each network's backward sweep (the trainer starts the all-reduce of finished buckets)


def grad_dst(param, shape, device):
    """Where a parameter-gradient kernel should write: the trainer's flat buffer slice, or a fresh tensor."""
    if GRAD_SINK is not None and param is not None:
        v = GRAD_SINK.get(id(param))
        if v is not None and v.numel() == int(torch.Size(shape).numel()):
            GRAD_SUNK.add(id(param))
            return v.view(shape)
    return torch.empty(shape, dtype=torch.float32, device=device)


def wino_dy_bn_transform(dt, yv, coef, mean, invstd):
    """BatchNorm backward (dt -> dy, in place) fused with the Winograd(tile 4) weight-gradient transform: -> Wt for
    conv2d_wgrad(wt_in=Wt); afterwards `dt` holds dy exactly as bn_bwd_apply would have left it."""
    L = lib()
    wt = torch.empty(L.fsd_wino_v_elems(yv.B, yv.H, yv.W, yv.C, 4), dtype=torch.float32, device=yv.t.device)
    check(L.fsd_wino_dy_bn_transform(dt.ptr, dt.ld, yv.ptr, yv.ld, coef.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                     wt.data_ptr(), yv.B, yv.H, yv.W, yv.C, 4, _stream()), "fsd_wino_dy_bn_transform")
    return wt


def wino_grad_transforms(dt, yv, coef, mean, invstd):
    """BatchNorm backward + both Winograd(tile 4) gradient transforms in one pass: -> (Vd, Wt) for
    conv3x3_wino(v_in=Vd, mode-1 weights) = data gradient and conv2d_wgrad(wt_in=Wt) = weight gradient."""
    L = lib()
    n = L.fsd_wino_v_elems(yv.B, yv.H, yv.W, yv.C, 4)
    vd = torch.empty(n, dtype=torch.float32, device=yv.t.device)
    wt = torch.empty(n, dtype=torch.float32, device=yv.t.device)
    check(L.fsd_wino_grad_transforms(dt.ptr, dt.ld, yv.ptr, yv.ld, coef.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                     vd.data_ptr(), wt.data_ptr(), yv.B, yv.H, yv.W, yv.C, 4, _stream()),
          "fsd_wino_grad_transforms")
    return vd, wt


def conv2d_wgrad(dyv, cout, xv, cin, ksize, dtype="f32", wino_v=None, param=None, tile=None, wt_in=None):
    """dW (cout, cin, k, k) from dy (View, columns [0,cout)) and the conv's NHWC input xv.
    wino_v: the forward pass's transformed input (conv3x3_wino keep_v, same tile), saves its recomputation.
    tile: 0 direct / 2 / 4 Winograd form (default: wino_tile's choice for this shape).
    wt_in: the already transformed gradient (wino_grad_transforms); dyv then only supplies the geometry.
    param: the parameter this is the gradient of (lets a trainer's gradient sink receive it directly)."""
    L = lib()
    dev = xv.t.device
    if xv.bf16 or dyv.bf16:
        return _conv2d_wgrad_h(dyv, cout, xv, cin, ksize, param)
    if tile is None:
        tile = wino_tile(cin, cout, ksize, xv.H, xv.W) if dtype == "f32" else 0
    if xv.lazy is not None and not (dtype == "f32" and ((tile and cin == xv.C and wino_v is not None)
                                                         or (not tile and ksize == 1 and lazy_ok_direct(xv, 1)))):
        raise ValueError("this weight gradient cannot form a deferred activation on load (materialise() x first)")
    if dtype == "f32" and tile and cin == xv.C:
        ws_bytes = L.fsd_wino_wgrad_workspace_bytes(xv.B, xv.H, xv.W, cin, cout, tile)
        ws = torch.empty((ws_bytes + 3) // 4, dtype=torch.float32, device=dev)
        dw = grad_dst(param, (cout, cin, 3, 3), dev)
        check(L.fsd_wino_conv3x3_wgrad(dyv.ptr, dyv.ld, xv.ptr, xv.ld, _ptr(wino_v), _ptr(wt_in), dw.data_ptr(),
                                       ws.data_ptr(), ws_bytes, xv.B, xv.H, xv.W, cin, cout, tile, _stream()),
              "fsd_wino_conv3x3_wgrad")
        return dw
    ws_bytes = L.fsd_conv2d_wgrad_workspace_bytes(xv.B, xv.H, xv.W, cin, cout, ksize)
    ws = torch.empty((ws_bytes + 3) // 4, dtype=torch.float32, device=dev)
    dw = grad_dst(param, (cout, cin, ksize, ksize), dev)
    if dtype != "f32":
        raise ValueError("fp32 views take the fp32 weight-gradient kernel (bf16 mode: bf16 views)")
    if xv.lazy is not None:
        sc, sh, sl = xv.lazy
        check(L.fsd_conv2d_wgrad_ex(dyv.ptr, dyv.ld, xv.ptr, xv.ld, dw.data_ptr(), ws.data_ptr(), ws_bytes, xv.B, xv.H,
                                    xv.W, cin, cout, ksize, sc.data_ptr(), sh.data_ptr(), float(sl), _stream()),
              "fsd_conv2d_wgrad_ex")
    else:
        check(L.fsd_conv2d_wgrad(dyv.ptr, dyv.ld, xv.ptr, xv.ld, dw.data_ptr(), ws.data_ptr(), ws_bytes, xv.B, xv.H,
                                 xv.W, cin, cout, ksize, _stream()), "fsd_conv2d_wgrad")
    return dw


def _conv2d_wgrad_h(dyv, cout, xv, cin, ksize, param):
    """bf16 storage mode: dW (float, OIHW) from bf16 dy and bf16 x."""
    _plain(xv, dyv)
    L = lib()
    dev = xv.t.device
    if not (xv.bf16 and dyv.bf16):
        raise ValueError("bf16 weight gradient needs bf16 dy and bf16 x")
    ws_bytes = L.fsd_conv2d_wgrad_h_workspace_bytes(xv.B, xv.H, xv.W, cin, cout, ksize)
    ws = torch.empty((ws_bytes + 3) // 4, dtype=torch.float32, device=dev)
    dw = grad_dst(param, (cout, cin, ksize, ksize), dev)
    check(L.fsd_conv2d_wgrad_h(dyv.ptr, dyv.ld, xv.ptr, xv.ld, dw.data_ptr(), ws.data_ptr(), ws_bytes, xv.B, xv.H, xv.W,
                               cin, cout, ksize, _stream()), "fsd_conv2d_wgrad_h")
    return dw


def c4_bnfused_eligible(xv, cout, ksize):
    """First-layer shape: NHWC4 input, 3x3, cout a multiple of 32 (darknet L0, reweighting-net L0)."""
    # the first-layer kernels address with 32-bit byte offsets: the (pixels, cout) activation must stay below 4 GiB
    fits = (xv.pixels + xv.W + 66) * max(4, cout) * 4 < 0xffffffff
    return ksize == 3 and xv.C == 4 and xv.ld == 4 and cout % 32 == 0 and xv.W >= 2 and fits


def conv3x3_wgrad_c4_bnfused(dt, yv, coef, mean, invstd, xv, cin, cout, param=None):
    """dW of a first layer straight from dt (gradient w.r.t. the BN output): BN backward fused, dy never stored."""
    L = lib()
    dev = xv.t.device
    ws_bytes = L.fsd_conv3x3_wgrad_c4_bnfused_workspace_bytes(xv.B, xv.H, xv.W, cout)
    ws = torch.empty((ws_bytes + 3) // 4, dtype=torch.float32, device=dev)
    dw = grad_dst(param, (cout, cin, 3, 3), dev)
    if dt.bf16 != yv.bf16:
        raise ValueError("dt and y must share the storage type")
    fn = L.fsd_conv3x3_wgrad_c4_bnfused_h if dt.bf16 else L.fsd_conv3x3_wgrad_c4_bnfused
    check(fn(dt.ptr, dt.ld, yv.ptr, yv.ld, coef.data_ptr(), mean.data_ptr(),
             invstd.data_ptr(), xv.ptr, xv.ld, dw.data_ptr(), ws.data_ptr(), ws_bytes,
             xv.B, xv.H, xv.W, cin, cout, _stream()), "fsd_conv3x3_wgrad_c4_bnfused")
    return dw


FUSE_FIRST_BWD = True     # one-sweep backward of a first conv block


def first_bwd_eligible(xv, yv, cout, ksize, pool, dz_full):
    """A first conv block the one-sweep backward takes: NHWC4 input, 3x3, BatchNorm, 2x2/2 max pool, even extents.
    Measured on the L0 shape (B = 64, 416x416, 32 channels, tools/probes/first_bwd_time.py): 0.91 ms against 1.28 ms for
    the unfused sequence in fp32, 0.94 against 1.07 ms in bf16 mode (the sweep is bound by its four 64-cycle fp32 MFMAs
    and ~60 VALU instructions per pooling cell, not by HBM, so halving the bytes buys it little)."""
    return (FUSE_FIRST_BWD and pool == 1 and dz_full is None and c4_bnfused_eligible(xv, cout, ksize)
            and xv.H % 2 == 0 and xv.W % 2 == 0 and yv.C == cout)


def first_layer_bwd(dz, yv, scale, shift, mean, invstd, slope, xv, cin, cout, bn, training, param=None):
    """-> (dW, dbeta, dgamma): the whole backward of a first conv block from dz (gradient w.r.t. its pooled output) in
    one sweep over y; dt is never written (csrc/first_bwd.hip)."""
    L = lib()
    dev = xv.t.device
    if dz.bf16 != yv.bf16:
        raise ValueError("dz and y must share the storage type")
    rows = L.fsd_first_layer_bwd_rows(xv.B, xv.H, xv.W)
    ws_bytes = L.fsd_first_layer_bwd_workspace_bytes(xv.B, xv.H, xv.W, cout)
    ws = torch.empty((ws_bytes + 3) // 4, dtype=torch.float32, device=dev)
    partial = torch.empty((rows, cout, 2), dtype=torch.float32, device=dev)
    fn = L.fsd_first_layer_bwd_accum_h if yv.bf16 else L.fsd_first_layer_bwd_accum
    check(fn(dz.ptr, dz.ld, yv.ptr, yv.ld, scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), invstd.data_ptr(), slope,
             xv.ptr, xv.ld, ws.data_ptr(), ws_bytes, partial.data_ptr(), xv.B, xv.H, xv.W, cin, cout, _stream()),
          "fsd_first_layer_bwd_accum")
    dbeta, dgamma, coef = reduce_partials(partial, yv.pixels, cout, scale=scale, want_coef=True, param0=bn.bias,
                                          param1=bn.weight)
    if not training:                     # frozen statistics: dy = scale * dt
        coef[1:].zero_()
    dw = grad_dst(param, (cout, cin, 3, 3), dev)
    check(L.fsd_first_layer_bwd_fold(ws.data_ptr(), ws_bytes, coef.data_ptr(), dw.data_ptr(), xv.B, xv.H, xv.W, cin, cout,
                                     _stream()), "fsd_first_layer_bwd_fold")
    return dw, dbeta, dgamma


# fp32 BatchNorm layers: the first backward pass only takes the statistics, the second one re-forms dt from the block-output
# gradient (bn_bwd_apply_g / wino_dy_bn_transform_g) -- dt is never written or read back.  DEFER_DT = False: the two-pass form.
DEFER_DT = True


def bn_act_pool_bwd(dz, dz_full, yv, scale, shift, mean, invstd, slope, pool, want_dt=True):
    """-> (dt View dense (pixels, C), partial [rows][C][2]).  want_dt=False (fp32): statistics only, dt is None."""
    L = lib()
    dev = yv.t.device
    dt = like_view(yv) if want_dt else None
    partial = torch.empty((L.fsd_bn_act_pool_bwd_rows(yv.B, yv.H, yv.W, pool), yv.C, 2), dtype=torch.float32,
                          device=dev)
    if dz.bf16 != yv.bf16 or (dz_full is not None and dz_full.bf16 != yv.bf16):
        raise ValueError("gradient and activation views must share the storage type")
    fn = L.fsd_bn_act_pool_bwd_h if yv.bf16 else L.fsd_bn_act_pool_bwd
    check(fn(dz.ptr, dz.ld, 0 if dz_full is None else dz_full.ptr,
             0 if dz_full is None else dz_full.ld, yv.ptr, yv.ld, _ptr(scale), _ptr(shift),
             _ptr(mean), _ptr(invstd), slope, pool, dt.ptr if want_dt else 0, partial.data_ptr(), yv.B, yv.H, yv.W,
             yv.C, _stream()), "fsd_bn_act_pool_bwd")
    return dt, partial


def bn_bwd_apply_g(dz, dz_full, yv, scale, shift, slope, pool, coef, mean, invstd):
    """Second pass after bn_act_pool_bwd(want_dt=False): -> dy View dense (pixels, C); bit-identical to
    bn_bwd_apply(bn_act_pool_bwd(...)[0], ...)."""
    dy = like_view(yv)
    fn = lib().fsd_bn_bwd_apply_g_h if yv.bf16 else lib().fsd_bn_bwd_apply_g
    check(fn(dz.ptr, dz.ld, 0 if dz_full is None else dz_full.ptr, 0 if dz_full is None else dz_full.ld,
             yv.ptr, yv.ld, _ptr(scale), _ptr(shift), slope, pool, coef.data_ptr(), mean.data_ptr(),
             invstd.data_ptr(), dy.ptr, yv.B, yv.H, yv.W, yv.C, _stream()), "fsd_bn_bwd_apply_g")
    return dy


def defer_dt_ok(yv, dz, dz_full):
    """Can the second pass re-form dt for these views?  fp32: always; bf16: 8-channel lanes (16-byte rows)."""
    if not yv.bf16:
        return True
    views = [yv, dz] + ([dz_full] if dz_full is not None else [])
    return yv.C % 8 == 0 and all(v.ld % 8 == 0 and v.c0 % 8 == 0 and v.t.data_ptr() % 16 == 0 for v in views)


def wino_dy_bn_transform_g(dz, dz_full, yv, scale, shift, slope, pool, coef, mean, invstd):
    """Second pass after bn_act_pool_bwd(want_dt=False) of a Winograd(tile 4) layer: -> (dy View, Wt); bit-identical to
    wino_dy_bn_transform on the dt of the two-pass form."""
    L = lib()
    dy = like_view(yv)
    wt = torch.empty(L.fsd_wino_v_elems(yv.B, yv.H, yv.W, yv.C, 4), dtype=torch.float32, device=yv.t.device)
    check(L.fsd_wino_dy_bn_transform_g(dz.ptr, dz.ld, 0 if dz_full is None else dz_full.ptr,
                                       0 if dz_full is None else dz_full.ld, yv.ptr, yv.ld, _ptr(scale), _ptr(shift), slope,
                                       pool, coef.data_ptr(), mean.data_ptr(), invstd.data_ptr(), dy.ptr, wt.data_ptr(),
                                       yv.B, yv.H, yv.W, yv.C, 4, _stream()), "fsd_wino_dy_bn_transform_g")
    return dy, wt


def reduce_partials(partial, count, channels, scale=None, want_coef=False, param0=None, param1=None):
    """-> (sum_col0 [C], sum_col1 [C], coef [3,C] or None).  param0 / param1: the parameters the two sums are the
    gradients of (BN bias / weight, or a conv bias), for the trainer's gradient sink."""
    L = lib()
    dev = partial.device
    s0, s1 = grad_dst(param0, (channels,), dev), grad_dst(param1, (channels,), dev)
    coef = torch.empty((3, channels), dtype=torch.float32, device=dev) if want_coef else None
    ws = torch.empty(L.fsd_reduce_workspace_bytes(channels) // 8, dtype=torch.float64, device=dev)
    check(L.fsd_bn_bwd_finalize(partial.data_ptr(), partial.shape[0], count, channels, _ptr(scale), s1.data_ptr(),
                                s0.data_ptr(), _ptr(coef), ws.data_ptr(), _stream()), "fsd_bn_bwd_finalize")
    return s0, s1, coef


def bn_bwd_apply(dt, yv, coef, mean, invstd):
    fn = lib().fsd_bn_bwd_apply_h if yv.bf16 else lib().fsd_bn_bwd_apply
    check(fn(dt.ptr, yv.ptr, yv.ld, coef.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
             yv.pixels, yv.C, _stream()), "fsd_bn_bwd_apply")
    return dt


def colsum(v, channels, param=None):
    L = lib()
    partial = torch.empty((L.fsd_act_bwd_rows(v.pixels), channels, 2), dtype=torch.float32, device=v.t.device)
    fn = L.fsd_colsum_partials_h if v.bf16 else L.fsd_colsum_partials
    check(fn(v.ptr, v.ld, partial.data_ptr(), v.pixels, channels, _stream()), "fsd_colsum_partials")
    s, _, _ = reduce_partials(partial, v.pixels, channels, param0=param)
    return s


def reorg_bwd(dout, xv, stride):
    dx = like_view(xv)
    fn = lib().fsd_reorg_bwd_h if xv.bf16 else lib().fsd_reorg_bwd
    check(fn(dout.ptr, dout.ld, dx.ptr, dx.ld, xv.B, xv.H, xv.W, xv.C, stride, _stream()), "fsd_reorg_bwd")
    return dx


def global_maxpool_bwd(dout, arg, xv):
    dx = like_view(xv)
    fn = lib().fsd_global_maxpool_bwd_h if xv.bf16 else lib().fsd_global_maxpool_bwd
    check(fn(dout.contiguous().data_ptr(), arg.data_ptr(), dx.ptr, dx.ld, xv.B, xv.H, xv.W, xv.C, _stream()),
          "fsd_global_maxpool_bwd")
    return dx


def add_inplace(dst, src):
    if dst.bf16 != src.bf16:
        raise ValueError("add_inplace: views must share the storage type")
    fn = lib().fsd_add_inplace_h if dst.bf16 else lib().fsd_add_inplace
    check(fn(dst.ptr, dst.ld, src.ptr, src.ld, dst.pixels, dst.C, _stream()), "fsd_add_inplace")
    return dst


def head_unfold_bwd(dweff, head_w, dyn, param=None):
    O, Cc = head_w.shape[0], head_w.shape[1]
    N = dyn.shape[0]
    d_head = grad_dst(param, (O, Cc, 1, 1), dyn.device)
    d_dyn = torch.empty((N, Cc, 1, 1), dtype=torch.float32, device=dyn.device)
    check(lib().fsd_head_unfold_bwd(dweff.data_ptr(), head_w.contiguous().data_ptr(), dyn.contiguous().data_ptr(),
                                    d_head.data_ptr(), d_dyn.data_ptr(), N, O, Cc, _stream()), "fsd_head_unfold_bwd")
    return d_head, d_dyn


def sgd_step(w, g, buf, lr, momentum, weight_decay, first):
    check(lib().fsd_sgd_step(w.data_ptr(), g.data_ptr(), buf.data_ptr(), lr, momentum, weight_decay,
                             1 if first else 0, w.numel(), _stream()), "fsd_sgd_step")


def sgd_multi_blocks(count, cout=0, cin=0, taps=0):
    return int(lib().fsd_sgd_multi_blocks(int(count), int(cout), int(cin), int(taps)))


def sgd_step_multi(w, g, buf, table, n_entries, total_blocks, elements, lr, momentum, weight_decay, first):
    """One launch of SGD(momentum, weight decay) over the tensors `table` lists inside the flat buffers, the bf16 operand copies
    of the conv weights re-packed from the updated values in the same pass (include/fsdet.h fsd_sgd_step_multi)."""
    check(lib().fsd_sgd_step_multi(w.data_ptr(), g.data_ptr(), buf.data_ptr(), table.data_ptr(), int(n_entries), int(total_blocks),
                                   int(elements), lr, momentum, weight_decay, 1 if first else 0, _stream()), "fsd_sgd_step_multi")


PROFILE_CLASSES = ("gemm_fwd", "gemm_wgrad", "wino_transform", "act_bwd", "act_fwd", "region", "sgd", "first_layer",
                   "gemm_bf16")


def kernel_profile(enable):
    """Switch the library's per-kernel-class HIP-event recording on / off (include/fsdet.h)."""
    lib().fsd_profile_enable(1 if enable else 0)


def kernel_profile_collect():
    """-> {class: dict(ms, work, launches)} summed over everything recorded since the last collect (synchronises)."""
    import ctypes as C
    n = lib().fsd_profile_num_classes()
    ms, work, cnt = (C.c_double * n)(), (C.c_double * n)(), (C.c_longlong * n)()
    check(lib().fsd_profile_collect(ms, work, cnt, n), "fsd_profile_collect")
    return {PROFILE_CLASSES[i]: dict(ms=ms[i], work=work[i], launches=int(cnt[i])) for i in range(n)}


def launch_count(reset=False):
    """Kernel launches the library has issued so far (since the last reset); include/fsdet.h fsd_launch_count."""
    return int(lib().fsd_launch_count(1 if reset else 0))


def clock_probe_mhz(device, iters=2000):
    """Shader clock (MHz) the GPU sustains under matrix-core load (a dependent-MFMA chain on every SIMD, ~1 ms)."""
    import ctypes as C
    scratch = torch.zeros(4, dtype=torch.float32, device=device)
    mhz = C.c_double(0.0)
    check(lib().fsd_clock_probe(scratch.data_ptr(), int(iters), C.byref(mhz), _stream()), "fsd_clock_probe")
    return float(mhz.value)



def f32_gemm_mode(mode=None):
    """Arithmetic of the fp32 GEMM kernels: "native" (fp32 MFMA) or "split" (six bf16 MFMA terms of three-way split
    operands, fp32-accurate; include/fsdet.h fsd_f32_gemm_mode).  Returns the previous mode; None only queries."""
    prev = lib().fsd_f32_gemm_mode(-1 if mode is None else {"native": 0, "split": 1}[mode])
    return "split" if prev else "native"
