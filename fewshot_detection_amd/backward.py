"""Reverse sweep over an engine tape: gradients of a cfg network (detector or reweighting net)
w.r.t. its parameters and its reweighting vectors, entirely on the HIP kernels.

Per fused conv block (reverse of engine.Network._conv):
    dz (+ dz_full) --fsd_bn_act_pool_bwd--> dt, partial sums     (maxpool argmax recomputed from y)
    partial sums   --fsd_bn_bwd_finalize--> dgamma, dbeta, coefficients
    dt             --fsd_bn_bwd_apply-----> dy                  (in place; first layer: fused into its weight gradient,
                                                                 fsd_conv3x3_wgrad_c4_bnfused; Winograd(4) layers: fused
                                                                 with both gradient transforms, fsd_wino_grad_transforms)
    dy, x          --fsd_conv2d_wgrad-----> dW                  (split-K MFMA GEMM over pixels)
    dy, W flipped  --fsd_conv2d_fwd-------> dx                  (same implicit-GEMM kernel as the forward)
"""
import torch

from . import ops, streams
from .ops import View

AVAILABLE = True
WGRAD_AFTER_DGRAD = __import__("os").environ.get("FSD_WGRAD_ORDER", "1") != "0"


def _accumulate(grads, view, g):
    key = id(view)
    if key in grads:
        ops.add_inplace(grads[key], g)
    else:
        grads[key] = g


def _wgrad_h(net, dy, cout, xv, cin, k, weight):
    """bf16 mode weight gradient; channel counts the transpose-read kernel does not take go through the fp32 kernel on
    the same bf16-valued operands (non-standard cfgs only)."""
    if dy.bf16 and xv.bf16 and cin % 8 == 0 and cout % 8 == 0 and xv.C == cin and dy.c0 % 8 == 0 and xv.c0 % 8 == 0:
        return ops.conv2d_wgrad(dy, cout, xv, cin, k, param=weight)
    net.fallback_convs += 1
    return ops.conv2d_wgrad(ops.cast_view(dy, torch.float32), cout, ops.cast_view(xv, torch.float32), cin, k, "f32", tile=0,
                            param=weight)


def _dgrad_h(net, dy, conv, xv, k):
    """bf16 mode data gradient (bf16 dy x bf16 rotated weights -> bf16 dx)."""
    if dy.bf16 and dy.C % 32 == 0 and xv.C % 2 == 0 and dy.c0 % 8 == 0:
        dx, _ = ops.conv2d(dy, net.cache.get(conv.weight, 1, "bf16"), xv.C, k)
        return dx
    net.fallback_convs += 1
    dyf = ops.cast_view(dy, torch.float32)
    if dyf.C % 4:
        pad = torch.zeros((dyf.pixels, (dyf.C + 3) // 4 * 4), dtype=torch.float32, device=dyf.t.device)
        pad[:, :dyf.C] = dyf.t
        dyf = View(pad, dy.B, dy.H, dy.W, pad.shape[1])
    wr = conv.weight.detach().to(torch.bfloat16).float()
    dxf, _ = ops.conv2d(dyf, ops.pack_weight(wr, 1, "f32"), xv.C, k)
    return ops.cast_view(dxf, torch.bfloat16)


def _off_path(ws, fn, reads):
    """Run the weight-gradient launch `fn` on the side stream `ws` (None: inline).  It starts once the main stream has
    produced its operands (`reads`: tensors allocated on the main stream that the side stream will still be reading
    after the caller has dropped them) and nothing on the main stream waits for it until the end of the sweep."""
    if ws is None:
        return fn()
    main = torch.cuda.current_stream()
    ws.wait_stream(main)
    with torch.cuda.stream(ws):
        dw = fn()
    streams.keep_alive(ws, *reads)
    streams.keep_alive(main, dw)
    return dw


def _conv_backward(net, rec, grads, pgrads, first_input, ws=None):
    gz = grads.pop(id(rec["z"]), None)
    gzf = grads.pop(id(rec["z_full"]), None) if rec.get("z_full") is not None else None
    if gz is None and gzf is None:
        return
    conv, bn, xv, k, cout = rec["conv"], rec["bn"], rec["x"], rec["k"], rec["cout"]
    cin = conv.weight.shape[1]
    wt_in = None
    direct = bn is None and rec["slope"] == 1.0 and rec["pool"] == 0
    if direct:
        dy = gz
        if conv.bias is not None:
            pgrads[id(conv.bias)] = ops.colsum(dy, cout, param=conv.bias)
    else:
        yv = rec["y"]
        pool = rec["pool"]
        if (bn is not None and xv is first_input and gz is not None
                and ops.first_bwd_eligible(xv, yv, cout, k, pool, gzf)):
            # first block of a network: no data gradient, so dt is only summed -- one sweep over y, dt never written
            def first():
                dw, dbeta, dgamma = ops.first_layer_bwd(gz, yv, rec["scale"], rec["shift"], rec["mean"], rec["invstd"],
                                                        rec["slope"], xv, cin, cout, bn, rec["training"], param=conv.weight)
                pgrads[id(bn.bias)], pgrads[id(bn.weight)] = dbeta, dgamma
                return dw
            pgrads[id(conv.weight)] = _off_path(ws, first, (gz.t, yv.t, rec["scale"], rec["shift"], rec["mean"],
                                                            rec["invstd"], xv.t))
            if ws is not None:
                streams.keep_alive(torch.cuda.current_stream(), pgrads[id(bn.bias)], pgrads[id(bn.weight)])
            return
        if gz is None:                       # only the un-pooled tap carries gradient
            gz, gzf, pool = gzf, None, 0
        kept = rec.get("wino_v")
        wino4 = bool(kept) and rec.get("wino_tile") == 4 and yv.C % 4 == 0
        # BatchNorm layers: dt is not materialised -- the first pass takes the statistics, the second one re-forms it
        defer = (ops.DEFER_DT and bn is not None and pool in (0, 1) and ops.defer_dt_ok(yv, gz, gzf)
                 and not (xv is first_input and ops.c4_bnfused_eligible(xv, cout, k))
                 and not (ops.FUSE_WINO_GRAD and wino4))
        dt, partial = ops.bn_act_pool_bwd(gz, gzf, yv, rec.get("scale"), rec.get("shift"), rec.get("mean"),
                                          rec.get("invstd"), rec["slope"], pool, want_dt=not defer)
        s1, s2, coef = ops.reduce_partials(partial, yv.pixels, cout, scale=rec.get("scale"), want_coef=bn is not None,
                                           param0=bn.bias if bn is not None else conv.bias,
                                           param1=bn.weight if bn is not None else None)
        if bn is not None:
            pgrads[id(bn.bias)] = s1
            pgrads[id(bn.weight)] = s2
            if not rec["training"]:          # frozen statistics: dy = scale * dt
                coef[1:].zero_()
            if xv is first_input and ops.c4_bnfused_eligible(xv, cout, k):
                # first layer: no data gradient is needed, so dy is formed inside the weight-gradient kernel
                pgrads[id(conv.weight)] = _off_path(
                    ws, lambda: ops.conv3x3_wgrad_c4_bnfused(dt, yv, coef, rec["mean"], rec["invstd"], xv, cin, cout,
                                                             param=conv.weight),
                    (dt.t, yv.t, coef, rec["mean"], rec["invstd"], xv.t))
                return
            kept = rec.get("wino_v")
            if (ops.FUSE_WINO_GRAD and kept and rec.get("wino_tile") == 4 and xv is not first_input and dt.C % 4 == 0
                    and ops.wino_tile(dt.C, xv.C, k, xv.H, xv.W) == 4):
                # Winograd(4) layer: BN backward + both gradient transforms in one pass over dt / y, dy never stored
                vd, wt = ops.wino_grad_transforms(dt, yv, coef, rec["mean"], rec["invstd"])
                pgrads[id(conv.weight)] = ops.conv2d_wgrad(dt, cout, xv, cin, k, "f32", wino_v=kept[0],
                                                           param=conv.weight, tile=4, wt_in=wt)
                dx, _ = ops.conv3x3_wino(dt, net.cache.get(conv.weight, 1, "wino4"), xv.C, tile=4, v_in=vd)
                _accumulate(grads, xv, dx)
                return
            if defer and wino4:
                dt, wt_in = ops.wino_dy_bn_transform_g(gz, gzf, yv, rec["scale"], rec["shift"], rec["slope"], pool, coef,
                                                       rec["mean"], rec["invstd"])
            elif defer:
                dt = ops.bn_bwd_apply_g(gz, gzf, yv, rec["scale"], rec["shift"], rec["slope"], pool, coef, rec["mean"],
                                        rec["invstd"])
            elif kept and rec.get("wino_tile") == 4 and dt.C % 4 == 0:
                # Winograd(4) layer: the BN backward rides on the weight-gradient transform (dt -> dy in place)
                wt_in = ops.wino_dy_bn_transform(dt, yv, coef, rec["mean"], rec["invstd"])
            else:
                ops.bn_bwd_apply(dt, yv, coef, rec["mean"], rec["invstd"])
        elif conv.bias is not None:
            pgrads[id(conv.bias)] = s1
        dy = dt
    kept = rec.get("wino_v")
    wtile = rec.get("wino_tile") or 0
    bf16 = net.compute_dtype == "bf16"
    # The weight gradient is off the critical path of the sweep: on the "wgrad" stream it runs beside the main stream's chain.
    # WHEN it is released matters: both the data gradient and the weight gradient of a layer are matrix-bound kernels that
    # fill every CU on their own -- launched side by side they only time-share the matrix pipes, and the HBM-bound kernels that
    # follow on the main stream (output transform, the next layer's statistics pass, gradient and input transforms) then run
    # with nothing beside them.  WGRAD_AFTER_DGRAD (default) queues the data gradient first and releases the weight gradient
    # behind it, so that it overlaps those HBM-bound kernels instead (FSD_WGRAD_ORDER=0: the round-2/3 order).
    def launch_wgrad():
        if bf16:
            pgrads[id(conv.weight)] = _off_path(ws, lambda: _wgrad_h(net, dy, cout, xv, cin, k, conv.weight), (dy.t, xv.t))
        else:
            pgrads[id(conv.weight)] = _off_path(
                ws, lambda: ops.conv2d_wgrad(dy, cout, xv, cin, k, "f32", wino_v=kept[0] if kept else None,
                                             param=conv.weight, tile=wtile, wt_in=wt_in),
                (dy.t, xv.t, kept[0] if kept else None, wt_in))
    after = WGRAD_AFTER_DGRAD and ws is not None
    if not after:
        launch_wgrad()
    if xv is not first_input and not rec.get("input_cast"):
        dyv = dy if dy.C % 4 == 0 else View(dy.t, dy.B, dy.H, dy.W, (dy.C + 3) // 4 * 4, dy.c0)
        tile = 0 if bf16 else ops.wino_tile(dyv.C, xv.C, k, xv.H, xv.W)
        if tile:
            dx, _ = ops.conv3x3_wino(dyv, net.cache.get(conv.weight, 1, "wino%d" % tile), xv.C, tile=tile)
        elif bf16:
            dx = _dgrad_h(net, dy, conv, xv, k)
        else:
            dx, _ = ops.conv2d(dyv, net.cache.get(conv.weight, 1, "f32"), xv.C, k)
        _accumulate(grads, xv, dx)
    if after:
        launch_wgrad()


def run_early(ctx, grad_out):
    """Backward sweep of the network behind autograd context `ctx` (darknet_meta._NetFn), run by its CONSUMER's sweep the
    moment d(output) exists.  The result is parked on the context; _NetFn.backward hands it to autograd when the engine
    gets there (and recomputes if autograd presents a different gradient, i.e. the output had another consumer)."""
    grad_out = grad_out.contiguous()
    if ctx.side is not None:
        main = torch.cuda.current_stream()
        s = streams.side(grad_out.device, ctx.side)
        s.wait_event(main.record_event())
        with torch.cuda.stream(s):
            grads = run(ctx.net, ctx.tape, grad_out, ctx.params, wgrad_stream=False)
        streams.keep_alive(s, grad_out)
    else:
        grads = run(ctx.net, ctx.tape, grad_out, ctx.params)
    # the gradient tensor itself is kept (not its bare address: the allocator could hand the address to another tensor)
    ctx.early_result = (grad_out, grads)
    ctx.tape = None          # consumed: the activations it kept are free for the next step (see _NetFn.backward)


def run(net, tape, grad_out, params, wgrad_stream=True):
    """Returns {"params": [grad or None, ... in the order of `params`], "dyn": grad of the vectors}.
    wgrad_stream: launch the weight gradients on the "wgrad" side stream (streams.py); the current stream waits for
    them before this returns."""
    ops.require_device(grad_out)
    ws = streams.side(grad_out.device, "wgrad") if (wgrad_stream and streams.ENABLED and streams.WGRAD) else None
    grads = {}
    pgrads = {}
    grad_dyn = None
    first_input = tape[0]["x"]
    for rec in reversed(tape):
        kind = rec["kind"]
        if kind == "output":
            x = rec["x"]
            g = ops.nchw_to_nhwc(grad_out.view(x.B, x.C, x.H, x.W), pad_to=4, dtype=x.t.dtype)
            grads[id(x)] = View(g.t, x.B, x.H, x.W, x.C, 0)
        elif kind == "head":
            x, head, dyn, n_cls, o_ch = rec["x"], rec["head"], rec["dyn"], rec["n_cls"], rec["o_ch"]
            rows = n_cls * o_ch
            w_eff = rec["w_eff"][:rows * x.C].view(rows, x.C, 1, 1)
            if net.compute_dtype == "bf16":
                # the loss gradient (float NCHW) becomes a bf16 NHWC matrix whose channel count is padded with zeros to a
                # multiple of 64: the reduction dimension of its data-gradient GEMM, a row count the weight gradient takes
                g = ops.nchw_to_nhwc(grad_out.view(x.B, rows, x.H, x.W), pad_to=64, dtype=torch.bfloat16)
                dx, _ = ops.conv2d(g, ops.pack_weight(w_eff, 1, "bf16"), x.C, 1)
                dweff = ops.conv2d_wgrad(g, g.C, x, x.C, 1)[:rows]
            else:
                g = ops.nchw_to_nhwc(grad_out.view(x.B, rows, x.H, x.W), pad_to=4)       # (B*HW, rows padded)
                dx, _ = ops.conv2d(g, ops.pack_weight(w_eff, 1, "f32"), x.C, 1)
                dweff = ops.conv2d_wgrad(g, rows, x, x.C, 1, "f32")
            _accumulate(grads, x, dx)
            d_head, d_dyn = ops.head_unfold_bwd(dweff, head.weight.detach(), dyn.detach(), param=head.weight)
            pgrads[id(head.weight)] = d_head
            grad_dyn = d_dyn
            early = streams.take_early(dyn)
            if early is not None and early.tape is not None:
                # the vectors came from the reweighting net of the same forward() call: queue ITS backward sweep now (on its
                # side stream when it has one) instead of after this whole sweep -- its gradients are complete a few ms
                # into the step, and a data-parallel trainer starts their all-reduce under the detector's backward
                run_early(early, d_dyn)
            # otherwise the reweighting net's backward (its own stream) may start as soon as this exists; published only when
            # the vectors really came from a network that ran on a side stream (nobody else would ever claim the event)
            elif streams.ENABLED and streams.META and streams.from_side(dyn):
                streams.publish(d_dyn, torch.cuda.current_stream().record_event())
            if head.bias is not None:
                dst = ops.grad_dst(head.bias, (o_ch,), g.t.device)
                torch.sum(ops.colsum(g, rows).view(n_cls, o_ch), dim=0, out=dst)
                pgrads[id(head.bias)] = dst
        elif kind == "globalmax":
            x = rec["x"]
            _accumulate(grads, x, ops.global_maxpool_bwd(grad_out.reshape(x.B, x.C), rec["arg"], x))
        elif kind == "globalavg":
            x = rec["x"]
            _accumulate(grads, x, ops.global_avgpool_bwd(grad_out.reshape(x.B, x.C), x))
        elif kind == "route":
            if len(rec["src"]) == 2:
                g = grads.pop(id(rec["z"]), None)
                if g is not None:
                    off = 0
                    for pv in rec["src_views"]:
                        _accumulate(grads, pv, View(g.t, g.B, g.H, g.W, pv.C, g.c0 + off))
                        off += pv.C
        elif kind == "reorg":
            g = grads.pop(id(rec["z"]), None)
            if g is not None:
                _accumulate(grads, rec["x"], ops.reorg_bwd(g, rec["x"], rec["stride"]))
        elif kind == "pool":
            g = grads.pop(id(rec["z"]), None)
            if g is not None:
                dt, _ = ops.bn_act_pool_bwd(g, None, rec["x"], None, None, None, None, 1.0, rec["pool"])
                _accumulate(grads, rec["x"], dt)
        elif kind == "conv":
            _conv_backward(net, rec, grads, pgrads, first_input, ws)
        elif kind == "input":
            pass
        else:
            raise NotImplementedError("backward of tape record %r" % kind)
        if ops.GRAD_HOOK is not None and kind in ("conv", "head"):
            ops.GRAD_HOOK((ws,))             # a trainer may start the all-reduce of a bucket this layer completed
    if ws is not None:
        torch.cuda.current_stream().wait_stream(ws)
    # gradients the kernels wrote straight into a trainer's flat buffer are not handed to autograd again
    return {"params": [None if id(p) in ops.GRAD_SUNK else pgrads.get(id(p)) for p in params], "dyn": grad_dyn}
