"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the C-ABI *contract* of
libmichigan_hip.so (include/michigan_hip.h), entry point by entry point.

Two uses, both from ``tests/`` only:
  * ``-m "not gpu"`` tests install it with ``michigan_amd._cabi.set_backend`` so the
    whole host stack (weight packing, dgrad parity decomposition, SPADE backward,
    module graphs, the gloo data-parallel path) runs here and is checked against
    the reference-derived oracle (oracle/michigan_oracle.py);
  * ``-m gpu`` tests run the real HIP kernels and this emulator on the same raw
    buffers and compare them (per-kernel contract parity).

It works on host memory through the same raw pointers the kernels get (ctypes
``from_address``), computes in float64 and rounds once to the storage dtype.  The
product never imports this module; michigan_amd has no CPU path.
"""
from __future__ import annotations

import ctypes
import math

import torch

MG_F32, MG_BF16 = 0, 1
ACT_NONE, ACT_RELU, ACT_LRELU, ACT_TANH = 0, 1, 2, 3
_TD = {MG_F32: torch.float32, MG_BF16: torch.bfloat16}


def _addr(p):
    if p is None:
        return 0
    if isinstance(p, ctypes.c_void_p):
        return p.value or 0
    return int(p)


def _view(p, shape, dtype):
    """A torch tensor aliasing host memory at raw address p."""
    addr = _addr(p)
    if addr == 0:
        return None
    n = 1
    for s in shape:
        n *= int(s)
    nbytes = n * torch.empty((), dtype=dtype).element_size()
    buf = (ctypes.c_char * nbytes).from_address(addr)
    return torch.frombuffer(buf, dtype=dtype, count=n).view(*[int(s) for s in shape])


def _act(v, act, slope):
    if act == ACT_RELU:
        return torch.clamp_min(v, 0)
    if act == ACT_LRELU:
        return torch.where(v > 0, v, v * slope)
    if act == ACT_TANH:
        return torch.tanh(v)
    return v


def _act_grad_from_out(y, act, slope):
    if act == ACT_RELU:
        return (y > 0).to(y.dtype)
    if act == ACT_LRELU:
        return torch.where(y > 0, torch.ones_like(y), torch.full_like(y, slope))
    if act == ACT_TANH:
        return 1 - y * y
    return torch.ones_like(y)


def _gather(x, hj, wj, isy, isx, dy, dx):
    """x [N,H,W,C] -> [N,hj,wj,C]: x[n, j*is + d] with zeros outside."""
    n, h, w, c = x.shape
    iy = torch.arange(hj) * isy + dy
    ix = torch.arange(wj) * isx + dx
    vy = (iy >= 0) & (iy < h)
    vx = (ix >= 0) & (ix < w)
    g = x[:, iy.clamp(0, h - 1)][:, :, ix.clamp(0, w - 1)]
    return g * (vy[:, None] & vx[None, :]).to(x.dtype)[None, :, :, None]


class EmulatorBackend:
    name = "emulator"

    # -- bookkeeping ---------------------------------------------------------
    def mg_abi_version(self):
        return 5

    def mg_wgrad_det_workspace(self, d):
        return 16            # the emulator's weight gradient is a deterministic float64 sum: nothing to size

    def mg_sizeof_desc(self, which):
        from michigan_amd import _cabi
        return ctypes.sizeof((_cabi.ConvDesc, _cabi.WgradDesc, _cabi.GradSlot, _cabi.PackJob, _cabi.SnLayer, _cabi.NormApply2Desc, _cabi.PyramidDesc)[which])

    def mg_last_error(self):
        return b""

    # -- conv ------------------------------------------------------------------
    def mg_conv_taps(self, d, stream=None):
        td = _TD[d.dtype]
        x = _view(d.in_, (d.N, d.Hin, d.Win, d.Cin), td).double()
        w = _view(d.wt, (d.ntaps, d.CoutP, d.Cin), td).double()
        out = _view(d.out, (d.N, d.Hout, d.Wout, d.Cout), td)
        acc = torch.zeros((d.N, d.Hj, d.Wj, d.Cout_gemm), dtype=torch.float64)
        for t in range(d.ntaps):
            g = _gather(x, d.Hj, d.Wj, d.isy, d.isx, int(d.tap_dy[t]), int(d.tap_dx[t]))
            acc += g @ w[t, :d.Cout_gemm].t()
        if _addr(d.bias):
            acc += _view(d.bias, (d.Cout_gemm,), torch.float32).double()
        oy = slice(d.ooy, d.ooy + (d.Hj - 1) * d.osy + 1, d.osy)
        ox = slice(d.oox, d.oox + (d.Wj - 1) * d.osx + 1, d.osx)
        if d.epilogue == 0:
            v = acc[..., :d.Cout]
            if _addr(d.resid):
                v = v + _view(d.resid, (d.N, d.Hout, d.Wout, d.Cout), td).double()[:, oy, ox]
            v = _act(v, d.act, d.slope)
            if _addr(d.x):                                   # ReLU / LeakyReLU-output mask of a data gradient (mask_slope 0 / slope)
                keep = _view(d.x, (d.N, d.Hout, d.Wout, d.Cout), td).double()[:, oy, ox] > 0
                v = torch.where(keep, v, v * float(d.mask_slope))
            out[:, oy, ox] = v.to(td)
        else:
            c = d.Cout
            ch = torch.arange(c)
            rg = 64 * (ch // 32) + ch % 32
            g1 = 1.0 + acc[..., rg]
            beta = acc[..., rg + 32]
            if d.x_up:                                       # x given as the half-resolution source of a nearest 2x upsample
                src = _view(d.x, (d.N, d.Hout // 2, d.Wout // 2, c), td).double()
                xs = src.repeat_interleave(2, 1).repeat_interleave(2, 2)[:, oy, ox]
            else:
                xs = _view(d.x, (d.N, d.Hout, d.Wout, c), td).double()[:, oy, ox]
            mean = _view(d.mean, (c,), torch.float32).double()
            rstd = _view(d.rstd, (c,), torch.float32).double()
            pre = (xs - mean) * rstd * g1 + beta
            out[:, oy, ox] = _act(pre, d.act, d.slope).to(td)
            if _addr(d.gamma_out):
                _view(d.gamma_out, (d.N, d.Hout, d.Wout, c), td)[:, oy, ox] = g1.to(td)
        return 0

    def mg_conv_wgrad(self, d, stream=None):
        td = _TD[d.dtype]
        x = _view(d.x, (d.N, d.Hin, d.Win, d.Cin), td).double()
        dy = _view(d.dy, (d.N, d.Hj, d.Wj, d.Cg), td).double().reshape(-1, d.Cg)
        dw = _view(d.dw, (d.ntaps, d.Cg, d.Cin), torch.float32)
        for t in range(d.ntaps):
            g = _gather(x, d.Hj, d.Wj, d.isy, d.isx, int(d.tap_dy[t]), int(d.tap_dx[t])).reshape(-1, d.Cin)
            dw[t] += (dy.t() @ g).float()
        if _addr(d.dbias):
            _view(d.dbias, (d.Cg,), torch.float32)[:] += dy.sum(0).float()
        return 0

    # -- statistics / norms ------------------------------------------------------
    def mg_stats_workspace(self, G, P, C):
        return 16

    def mg_channel_stats(self, x, dtype, G, P, C, shift, sums, partial, stream=None):
        xv = _view(x, (G, P, C), _TD[dtype]).double()
        s = _view(sums, (G, 2, C), torch.float64)
        s[:, 0] = xv.sum(1)
        s[:, 1] = (xv * xv).sum(1)
        return 0

    def mg_channel_stats_finalize(self, x, dtype, G, P, C, sum_scale, count, eps, momentum, running_mean, running_var, sums, mean, rstd,
                                  partial, stream=None):
        self.mg_channel_stats(x, dtype, G, P, C, 1, sums, partial)
        _view(sums, (G, 2, C), torch.float64).mul_(sum_scale)
        return self.mg_norm_finalize(sums, G, C, count, eps, momentum, running_mean, running_var, mean, rstd)

    def mg_norm_finalize(self, sums, G, C, count, eps, momentum, running_mean, running_var, mean, rstd, stream=None):
        s = _view(sums, (G, 2, C), torch.float64)
        m = s[:, 0] / count
        var = (s[:, 1] / count - m * m).clamp_min(0)
        _view(mean, (G, C), torch.float32)[:] = m.float()
        _view(rstd, (G, C), torch.float32)[:] = torch.rsqrt(var + eps).float()
        if _addr(running_mean):
            rm, rv = _view(running_mean, (C,), torch.float32), _view(running_var, (C,), torch.float32)
            rm.mul_(1 - momentum).add_(m[0].float(), alpha=momentum)
            rv.mul_(1 - momentum).add_((var[0] * (count / max(count - 1, 1))).float(), alpha=momentum)
        return 0

    def mg_norm_act_fwd(self, x, y, dtype, G, P, C, mean, rstd, act, slope, resid=None, stream=None):
        td = _TD[dtype]
        xv = _view(x, (G, P, C), td).double()
        mu = _view(mean, (G, 1, C), torch.float32).double()
        rs = _view(rstd, (G, 1, C), torch.float32).double()
        out = _act((xv - mu) * rs, act, slope)
        if _addr(resid):
            out = out + _view(resid, (G, P, C), td).double()
        _view(y, (G, P, C), td)[:] = out.to(td)
        return 0

    def _bwd_common(self, dh, h, x, g1, dtype, G, P, C, mean, rstd, act, slope):
        td = _TD[dtype]
        dhv = _view(dh, (G, P, C), td).double()
        hv = _view(h, (G, P, C), td).double() if _addr(h) else torch.ones_like(dhv)    # optional when act == none
        xv = _view(x, (G, P, C), td).double()
        mu = _view(mean, (G, 1, C), torch.float32).double()
        rs = _view(rstd, (G, 1, C), torch.float32).double()
        dpre = dhv * _act_grad_from_out(hv, act, slope)
        xh = (xv - mu) * rs
        g1v = _view(g1, (G, P, C), td)
        dxh = dpre * g1v.double() if g1v is not None else dpre
        return dpre, xh, dxh, rs

    def mg_norm_bwd_reduce(self, dh, h, x, g1, dtype, G, P, C, mean, rstd, act, slope, dgb, sums, partial, stream=None):
        dpre, xh, dxh, _ = self._bwd_common(dh, h, x, g1, dtype, G, P, C, mean, rstd, act, slope)
        s = _view(sums, (G, 2, C), torch.float32)
        s[:, 0] = dxh.sum(1).float()
        s[:, 1] = (dxh * xh).sum(1).float()
        if _addr(dgb):
            td = _TD[dtype]
            cr2 = 2 * ((C + 31) // 32) * 32
            out = _view(dgb, (P, cr2), td)
            ch = torch.arange(C)
            rg = 64 * (ch // 32) + ch % 32
            out[:, rg] = (dpre[0] * xh[0]).to(td)
            out[:, rg + 32] = dpre[0].to(td)
        return 0

    def mg_norm_bwd_reduce_up(self, dh, h, x, g1, dtype, N, H, W, C, mean, rstd, act, slope, dgb, sums, partial, stream=None):
        td = _TD[dtype]
        xf = _view(x, (N, H // 2, W // 2, C), td).repeat_interleave(2, 1).repeat_interleave(2, 2).contiguous()
        return self.mg_norm_bwd_reduce(dh, h, xf.data_ptr(), g1, dtype, 1, N * H * W, C, mean, rstd, act, slope, dgb, sums, partial)

    def mg_norm_apply2_supported(self, dtype, C):
        vec = 8 if dtype == MG_BF16 else 4
        return int(C % vec == 0 and C // vec <= 256 and 256 % (C // vec) == 0)

    def mg_norm_bwd_apply2(self, d, stream=None):
        """Contract of mg_norm_bwd_apply2 (include/michigan_hip.h): float64, one rounding."""
        td, C, P = _TD[d.dtype], d.C, d.P
        if d.up:
            n = P // (d.H * d.W)
            xs = _view(d.x, (n, d.H // 2, d.W // 2, C), td).double()
            xfull = xs.repeat_interleave(2, 1).repeat_interleave(2, 2).reshape(P, C)
        else:
            xfull = _view(d.x, (P, C), td).double()
        mean = _view(d.mean, (C,), torch.float32).double()
        rs = _view(d.rstd, (C,), torch.float32).double()
        xh = (xfull - mean) * rs
        total = torch.zeros((P, C), dtype=torch.float64)
        for b in range(2):
            if not d.dh[b]:
                continue
            dxh = _view(d.dh[b], (P, C), td).double()
            if d.h[b]:
                dxh = dxh * _act_grad_from_out(_view(d.h[b], (P, C), td).double(), d.act[b], d.slope[b])
            if d.g1[b]:
                dxh = dxh * _view(d.g1[b], (P, C), td).double()
            sums = _view(d.sums[b], (2, C), torch.float32).double() * float(d.inv_count)
            total += rs * (dxh - sums[0] - xh * sums[1])
        if d.up:
            total = total.view(n, d.H // 2, 2, d.W // 2, 2, C).sum((2, 4))
            _view(d.dx, (n, d.H // 2, d.W // 2, C), td)[:] = total.to(td)
        else:
            _view(d.dx, (P, C), td)[:] = total.to(td)
        return 0

    def mg_norm_bwd_apply(self, dh, h, x, g1, dtype, G, P, C, mean, rstd, s1, s2, sum_gstride, sum_scale, act, slope, dx, stream=None):
        _, xh, dxh, rs = self._bwd_common(dh, h, x, g1, dtype, G, P, C, mean, rstd, act, slope)
        a = torch.stack([_view(_addr(s1) + 4 * g * sum_gstride, (C,), torch.float32) for g in range(G)]).double().view(G, 1, C) * sum_scale
        b = torch.stack([_view(_addr(s2) + 4 * g * sum_gstride, (C,), torch.float32) for g in range(G)]).double().view(G, 1, C) * sum_scale
        td = _TD[dtype]
        _view(dx, (G, P, C), td)[:] = (rs * (dxh - a - xh * b)).to(td)
        return 0

    def mg_act_bwd(self, dy, y, dpre, dtype, numel, act, slope, stream=None):
        td = _TD[dtype]
        d = _view(dy, (numel,), td).double()
        v = _view(y, (numel,), td).double()
        _view(dpre, (numel,), td)[:] = (d * _act_grad_from_out(v, act, slope)).to(td)
        return 0

    # -- resampling ----------------------------------------------------------------
    def mg_upsample2x_fwd(self, x, y, dtype, N, H, W, C, stream=None):
        td = _TD[dtype]
        xv = _view(x, (N, H, W, C), td)
        _view(y, (N, 2 * H, 2 * W, C), td)[:] = xv.repeat_interleave(2, 1).repeat_interleave(2, 2)
        return 0

    def mg_upsample2x_bwd(self, dy, dx, dtype, N, H, W, C, stream=None):
        td = _TD[dtype]
        d = _view(dy, (N, H, 2, W, 2, C), td).double()
        _view(dx, (N, H, W, C), td)[:] = d.sum((2, 4)).to(td)
        return 0

    def mg_reflect_pad_fwd(self, x, y, dtype, N, H, W, C, P, stream=None):
        td = _TD[dtype]
        xv = _view(x, (N, H, W, C), td).permute(0, 3, 1, 2)
        out = torch.nn.functional.pad(xv.double(), (P, P, P, P), mode="reflect").permute(0, 2, 3, 1)
        _view(y, (N, H + 2 * P, W + 2 * P, C), td)[:] = out.to(td)
        return 0

    def mg_reflect_pad_bwd(self, dy, dx, dtype, N, H, W, C, P, stream=None):
        td = _TD[dtype]
        d = _view(dy, (N, H + 2 * P, W + 2 * P, C), td).double()
        iy = (torch.arange(-P, H + P).abs()); iy = torch.where(iy >= H, 2 * H - 2 - iy, iy)
        ix = (torch.arange(-P, W + P).abs()); ix = torch.where(ix >= W, 2 * W - 2 - ix, ix)
        acc = torch.zeros(N, H, W + 2 * P, C, dtype=torch.float64).index_add_(1, iy, d)
        out = torch.zeros(N, H, W, C, dtype=torch.float64).index_add_(2, ix, acc)
        _view(dx, (N, H, W, C), td)[:] = out.to(td)
        return 0

    @staticmethod
    def _pool_cnt(L, Lo):
        o = torch.arange(Lo)
        lo = (2 * o - 1).clamp_min(0)
        hi = (2 * o + 1).clamp_max(L - 1)
        return (hi - lo + 1).double()

    def mg_avgpool3s2_fwd(self, x, y, dtype, N, H, W, C, stream=None):
        td = _TD[dtype]
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        xv = _view(x, (N, H, W, C), td).double()
        s = torch.zeros((N, Ho, Wo, C), dtype=torch.float64)
        for ky in (-1, 0, 1):
            for kx in (-1, 0, 1):
                s += _gather(xv, Ho, Wo, 2, 2, ky, kx)
        cnt = self._pool_cnt(H, Ho)[:, None] * self._pool_cnt(W, Wo)[None, :]
        _view(y, (N, Ho, Wo, C), td)[:] = (s / cnt[None, :, :, None]).to(td)
        return 0

    def mg_avgpool3s2_bwd(self, dy, dx, dtype, N, H, W, C, stream=None):
        td = _TD[dtype]
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        d = _view(dy, (N, Ho, Wo, C), td).double()
        cnt = self._pool_cnt(H, Ho)[:, None] * self._pool_cnt(W, Wo)[None, :]
        d = d / cnt[None, :, :, None]
        out = torch.zeros((N, H + 2, W + 2, C), dtype=torch.float64)
        for ky in (-1, 0, 1):
            for kx in (-1, 0, 1):
                out[:, 1 + ky:1 + ky + 2 * Ho:2, 1 + kx:1 + kx + 2 * Wo:2][:, :Ho, :Wo] += d
        _view(dx, (N, H, W, C), td)[:] = out[:, 1:H + 1, 1:W + 1].to(td)
        return 0

    def mg_maxpool2_fwd(self, x, y, dtype, N, H, W, C, stream=None):
        td = _TD[dtype]
        Ho, Wo = H // 2, W // 2
        xv = _view(x, (N, H, W, C), td)[:, :2 * Ho, :2 * Wo].reshape(N, Ho, 2, Wo, 2, C)
        _view(y, (N, Ho, Wo, C), td)[:] = xv.amax((2, 4))
        return 0

    def mg_assemble_nhwc8(self, planar, cp, nhwc, cs, cf, out, dtype, N, HW, stream=None):
        td = _TD[dtype]
        o = _view(out, (N, HW, 8), td)
        o.zero_()
        if cp:
            o[:, :, :cp] = _view(planar, (N, cp, HW), torch.float32).permute(0, 2, 1).to(td)
        if cf:
            o[:, :, cp:cp + cf] = _view(nhwc, (N, HW, cs), td)[:, :, :cf]
        return 0

    def mg_grad_sum_act(self, g1, g2, y, out, dtype, numel, act, slope, stream=None):
        td = _TD[dtype]
        g = _view(g1, (numel,), td).double()
        if _addr(g2):
            g = g + _view(g2, (numel,), td).double()
        _view(out, (numel,), td)[:] = (g * _act_grad_from_out(_view(y, (numel,), td).double(), act, slope)).to(td)
        return 0

    def mg_maxpool2_bwd(self, dy, x, dx, dtype, N, H, W, C, relu_input=0, stream=None):
        td = _TD[dtype]
        Ho, Wo = H // 2, W // 2
        xv = _view(x, (N, H, W, C), td).float()[:, :2 * Ho, :2 * Wo].reshape(N, Ho, 2, Wo, 2, C)
        win = xv.permute(0, 1, 3, 5, 2, 4).reshape(N, Ho, Wo, C, 4)
        # first maximum in window scan order (ATen max_pool2d picks the first with val > max)
        arg = torch.zeros((N, Ho, Wo, C), dtype=torch.long)
        best = win[..., 0].clone()
        for k in range(1, 4):
            better = win[..., k] > best
            best = torch.where(better, win[..., k], best)
            arg = torch.where(better, torch.full_like(arg, k), arg)
        d = _view(dy, (N, Ho, Wo, C), td).float()
        g = torch.zeros((N, Ho, Wo, C, 4))
        g.scatter_(4, arg[..., None], d[..., None])
        full = torch.zeros((N, H, W, C))
        full[:, :2 * Ho, :2 * Wo] = g.reshape(N, Ho, Wo, C, 2, 2).permute(0, 1, 4, 2, 5, 3).reshape(N, 2 * Ho, 2 * Wo, C)
        if relu_input:
            full = full * (_view(x, (N, H, W, C), td).float() > 0)
        _view(dx, (N, H, W, C), td)[:] = full.to(td)
        return 0

    def mg_blend_fwd(self, bg, x, hair, back, y, dtype, P, C, act=0, slope=0.2, stream=None):
        td = _TD[dtype]
        b = _view(bg, (P, C), td).double()
        v = _view(x, (P, C), td).double()
        hm = _view(hair, (P, 1), torch.float32).double()
        bm = _view(back, (P, 1), torch.float32).double()
        _view(y, (P, C), td)[:] = _act(b * (1 - hm) + v * (1 - bm), act, slope).to(td)
        return 0

    def mg_blend_bwd(self, dy, y, hair, back, dbg, dx, dtype, P, C, act=0, slope=0.2, stream=None):
        td = _TD[dtype]
        d = _view(dy, (P, C), td).double()
        if act != ACT_NONE:
            d = d * _act_grad_from_out(_view(y, (P, C), td).double(), act, slope)
        hm = _view(hair, (P, 1), torch.float32).double()
        bm = _view(back, (P, 1), torch.float32).double()
        if _addr(dbg):
            _view(dbg, (P, C), td)[:] = (d * (1 - hm)).to(td)
        if _addr(dx):
            _view(dx, (P, C), td)[:] = (d * (1 - bm)).to(td)
        return 0

    @staticmethod
    def _gemm_rows(cout, two):
        co = torch.arange(cout)
        return (64 * (co // 32) + co % 32) if two else co

    def mg_pack_weight(self, w0, w1, dst, dtype, cout, cin, taps, rows_p, cols_p, mode, stream=None):
        td = _TD[dtype]
        two = _addr(w1) != 0
        out = _view(dst, (taps, rows_p, cols_p), td)
        out.zero_()
        rows = self._gemm_rows(cout, two)
        for which, p in enumerate((w0, w1) if two else (w0,)):
            w = _view(p, (cout, cin, taps), torch.float32).permute(2, 0, 1)        # [t, co, ci]
            r = rows + 32 * which
            if mode == 0:
                out[:, r, :cin] = w.to(td)
            else:
                out[:, :cin, r] = w.permute(0, 2, 1).to(td)
        return 0

    def mg_unpack_wgrad(self, dw, d0, d1, cout, cin, taps, rows, cols, stream=None):
        src = _view(dw, (taps, rows, cols), torch.float32)
        two = _addr(d1) != 0
        r = self._gemm_rows(cout, two)
        for which, p in enumerate((d0, d1) if two else (d0,)):
            _view(p, (cout, cin, taps), torch.float32)[:] = src[:, r + 32 * which, :cin].permute(1, 2, 0)
        return 0

    def mg_pack_job_blocks(self, mode, cout, cin, taps, rows_p, cols_p):
        if mode == 2:
            return (cout * cin * taps + 1023) // 1024
        return -1 if taps > 49 else ((rows_p + 3) // 4) * ((cols_p + 63) // 64)

    def mg_pack_weights(self, jobs, njobs, block_job, nblocks, stream=None):
        from michigan_amd import _cabi
        for j in (_cabi.PackJob * njobs).from_address(_addr(jobs)):
            sig = float(_view(j.sigma, (1,), torch.float32)[0]) if j.sigma else None
            if j.mode == 2:
                w = _view(j.w0, (j.cout * j.cin * j.taps,), torch.float32)
                _view(j.dst, (j.cout * j.cin * j.taps,), torch.float32)[:] = w / sig if sig is not None else w
                continue
            srcs = []
            for p in (j.w0, j.w1):
                if p:
                    w = _view(p, (j.cout, j.cin, j.taps), torch.float32)
                    srcs.append((w / sig if sig is not None else w).contiguous())
            self.mg_pack_weight(srcs[0].data_ptr(), srcs[1].data_ptr() if len(srcs) > 1 else None, j.dst, j.dtype, j.cout, j.cin,
                                j.taps, j.rows_p, j.cols_p, j.mode)
        return 0

    def mg_sn_layer_blocks(self, rows, cols, which):
        return ((cols + 255) // 256) * ((rows + 31) // 32) if which == 0 else ((rows + 3) // 4 if which == 1 else (rows + 31) // 32)

    def mg_sn_power_iteration(self, layers, nlayers, bl1, nb1, bl3, nb3, do_power_iteration, eps, stream=None):
        """torch/nn/utils/spectral_norm.py compute_weight (dim 0, one iteration), layer by layer, in float64."""
        from michigan_amd import _cabi
        for L in (_cabi.SnLayer * nlayers).from_address(_addr(layers)):
            w = _view(L.w, (L.rows, L.cols), torch.float32).double()
            u, v = _view(L.u, (L.rows,), torch.float32), _view(L.v, (L.cols,), torch.float32)
            if do_power_iteration:
                t1 = w.t() @ u.double()
                vn = t1 / max(float(t1.norm()), eps)
                v[:] = vn.float()
                if L.v_copy:
                    _view(L.v_copy, (L.cols,), torch.float32)[:] = vn.float()
                t2 = w @ v.double()
                un = t2 / max(float(t2.norm()), eps)
                u[:] = un.float()
                if L.u_copy:
                    _view(L.u_copy, (L.rows,), torch.float32)[:] = un.float()
            else:
                t2 = w @ v.double()
            _view(L.sigma, (1,), torch.float32)[0] = float(u.double() @ t2)
        return 0

    def mg_grad_slot_blocks(self, cout, cin, ntens):
        return ntens * ((cin + 63) // 64) * ((cout + 3) // 4) + 1

    def mg_grad_drain(self, table, nslots, block_slot, nblocks, partial, stream=None):
        """Contract of include/michigan_hip.h `mg_grad_drain`, slot by slot (float64 arithmetic, one rounding)."""
        from michigan_amd import _cabi
        slots = (_cabi.GradSlot * nslots).from_address(_addr(table))
        for sl in slots:
            two = bool(sl.dst1)
            shape = (sl.taps, sl.cols, sl.rows) if sl.swapped else (sl.taps, sl.rows, sl.cols)
            g = _view(sl.gemm, shape, torch.float32)
            gg = g.permute(0, 2, 1) if sl.swapped else g                      # [t, row, ci]
            r = self._gemm_rows(sl.cout, two)
            for which, dst in enumerate((sl.dst0, sl.dst1) if two else (sl.dst0,)):
                val = gg[:, r + 32 * which, :sl.cin].permute(1, 2, 0).double()       # [co, ci, t]
                if sl.w_sn:
                    w_sn = _view(sl.w_sn, (sl.cout, sl.cin, sl.taps), torch.float32).double()
                    u = _view(sl.u, (sl.cout,), torch.float32).double()
                    v = _view(sl.v, (sl.cin * sl.taps,), torch.float32).double().view(sl.cin, sl.taps)
                    sigma = float(_view(sl.sigma, (1,), torch.float32)[0])
                    sdot = (val * w_sn).sum()
                    val = (val - sdot * u[:, None, None] * v[None]) / sigma
                d = _view(dst, (sl.cout, sl.cin, sl.taps), torch.float32)
                d += val.float()
            if sl.dbias_gemm:
                nrows = int(r.max()) + 33 if two else sl.cout
                db = _view(sl.dbias_gemm, (nrows,), torch.float32)
                for which, dst in enumerate((sl.dbias0, sl.dbias1) if two else (sl.dbias0,)):
                    _view(dst, (sl.cout,), torch.float32)[:] += db[r + 32 * which]
                db.zero_()
            g.zero_()
        return 0

    def mg_l1_mean_fwd(self, a, b, dtype, numel, out, partial, stream=None):
        td = _TD[dtype]
        av, bv = _view(a, (numel,), td).double(), _view(b, (numel,), td).double()
        _view(out, (1,), torch.float32)[0] = float((av - bv).abs().mean())
        return 0

    def mg_l1_mean_bwd(self, a, b, gscale, dtype, numel, da, stream=None):
        td = _TD[dtype]
        av, bv = _view(a, (numel,), td).double(), _view(b, (numel,), td).double()
        g = float(_view(gscale, (1,), torch.float32)[0]) / numel
        _view(da, (numel,), td)[:] = (torch.sign(av - bv) * g).to(td)
        return 0

    def mg_wide_edge_weight(self, label, N, Hl, Wl, h, w, k, wide, out, stream=None):
        """loss.py:60-80 with torch's own ops (float32 like the reference: F.interpolate's index arithmetic is float32)."""
        import torch.nn.functional as F
        lab = _view(label, (N, 1, Hl, Wl), torch.float32)
        t = F.interpolate(lab, size=(h, w), mode="nearest")
        p = int(k / 2)
        grown = F.max_pool2d(t, kernel_size=k, stride=1, padding=p)
        shrunk = 1 - F.max_pool2d(1 - t, kernel_size=k, stride=1, padding=p)
        e = F.interpolate(grown - shrunk, size=(h, w), mode="nearest")
        _view(out, (N, 1, h, w), torch.float32)[:] = e * wide + (1 - e)
        return 0

    @staticmethod
    def _hinge_term(x, mode):
        if mode == 0:
            return x
        return torch.clamp_max((x if mode == 1 else -x) - 1, 0)

    def mg_hinge_fwd(self, x, weight, dtype, n, mode, out, stream=None):
        v = self._hinge_term(_view(x, (n,), _TD[dtype]).double(), mode)
        if _addr(weight):
            v = v * _view(weight, (n,), torch.float32).double()
        _view(out, (1,), torch.float32)[0] = float(-v.mean())
        return 0

    def mg_hinge_bwd(self, x, weight, g, dtype, n, mode, dx, stream=None):
        xv = _view(x, (n,), _TD[dtype]).double()
        if mode == 0:
            d = torch.ones_like(xv)
        elif mode == 1:
            d = (xv - 1 < 0).double()
        else:
            d = -((-xv - 1) < 0).double()
        if _addr(weight):
            d = d * _view(weight, (n,), torch.float32).double()
        _view(dx, (n,), _TD[dtype])[:] = (-float(_view(g, (1,), torch.float32)[0]) / n * d).to(_TD[dtype])
        return 0

    @staticmethod
    def _gray(img):
        x = (img[..., :3].double() + 1) / 2 * 255
        return 0.299 * x[..., 0] + 0.587 * x[..., 1] + 0.144 * x[..., 2]           # [N,H,W]

    def mg_gabor_argmax_fwd(self, img, bank, conf, idx, dtype, N, H, W, C, stream=None):
        import torch.nn.functional as F
        g = self._gray(_view(img, (N, H, W, C), _TD[dtype]))[:, None]
        k = _view(bank, (32, 1, 17, 17), torch.float32).double()
        r = F.conv2d(g, k, padding=8).clamp_min(0)
        _view(conf, (N, H, W), torch.float32)[:] = r.max(1)[0].float()
        _view(idx, (N, H, W), torch.uint8)[:] = r.argmax(1).to(torch.uint8)
        return 0

    def mg_gabor_argmax_bwd(self, dconf, idx, bank, dimg, dtype, N, H, W, C, stream=None):
        import torch.nn.functional as F
        td = _TD[dtype]
        g = _view(dconf, (N, H, W), torch.float32).double()
        ix = _view(idx, (N, H, W), torch.uint8).long()
        k = _view(bank, (32, 1, 17, 17), torch.float32).double()
        onehot = torch.zeros((N, 32, H, W), dtype=torch.float64).scatter_(1, ix[:, None], g[:, None])
        dgray = F.conv_transpose2d(onehot, k, padding=8)[:, 0]                       # adjoint of the correlation
        out = _view(dimg, (N, H, W, C), td)
        out.zero_()
        for c, wgt in enumerate((0.299, 0.587, 0.144)):
            out[..., c] = (dgray * wgt * 127.5).to(td)
        return 0

    def mg_self_attention(self, q, k, v, out, dtype, N, L, d_qk, d_v, ldq, ldk, ldv, ldo, stream=None):
        """Contract of mg_self_attention (generator.py:467-485): float64 softmax(q k^T) v, one rounding to the storage dtype."""
        td = _TD[dtype]
        es = torch.empty((), dtype=td).element_size()

        def rows(p, ld, width):            # [N, L, width] strided view of rows `ld` elements apart
            flat = _view(p, ((N * L - 1) * ld + width,), td)
            return flat.as_strided((N, L, width), (L * ld, ld, 1))
        qv, kv, vv = rows(q, ldq, d_qk).double(), rows(k, ldk, d_qk).double(), rows(v, ldv, d_v).double()
        res = torch.softmax(qv @ kv.transpose(1, 2), dim=-1) @ vv                          # [N, L, d_v]
        rows(out, ldo, d_v)[:] = res.to(td)
        return 0

    def mg_sn_normalize(self, t, n, eps, dst, dst2, sigma, stream=None):
        tv = _view(t, (n,), torch.float32).double()
        q = tv / max(tv.norm().item(), eps)
        _view(dst, (n,), torch.float32)[:] = q.float()
        if _addr(dst2):
            _view(dst2, (n,), torch.float32)[:] = q.float()
        if _addr(sigma):
            _view(sigma, (1,), torch.float32)[0] = float((q.float().double() * tv).sum())
        return 0

    def mg_sn_scale(self, w, sigma, out, numel, stream=None):
        s = _view(sigma, (1,), torch.float32)[0]
        _view(out, (numel,), torch.float32)[:] = _view(w, (numel,), torch.float32) / s
        return 0

    def mg_sn_bwd(self, g, u, v, s, sigma, out, rows, cols, stream=None):
        gv = _view(g, (rows, cols), torch.float32).double()
        uv = torch.outer(_view(u, (rows,), torch.float32).double(), _view(v, (cols,), torch.float32).double())
        sv, sg = _view(s, (1,), torch.float32)[0].double(), _view(sigma, (1,), torch.float32)[0].double()
        _view(out, (rows, cols), torch.float32)[:] = ((gv - sv * uv) / sg).float()
        return 0

    def mg_set_option(self, key, value):
        return 0

    def mg_adam_step(self, param, grad, m, v, numel, lr, b1, b2, eps, step, gscale, stream=None):
        p = _view(param, (numel,), torch.float32)
        g = _view(grad, (numel,), torch.float32) * gscale
        mm = _view(m, (numel,), torch.float32)
        vv = _view(v, (numel,), torch.float32)
        mm.lerp_(g, 1 - b1)
        vv.mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1 = 1 - b1 ** step
        bc2 = 1 - b2 ** step
        denom = vv.sqrt() / math.sqrt(bc2) + eps
        p.addcdiv_(mm, denom, value=-lr / bc1)
        return 0

    # -- input pipeline: the contract is the reference restatement in oracle/inputs_oracle.py --
    def mg_input_crop_u8(self, src, dst, crop, ytab, xtab, mul, N, Hs, Ws, C, H, W, mode, unknown_label, stream=None):
        from oracle import inputs_oracle as IO
        cr = _view(crop, (N, 3), torch.int32).numpy()
        yt = None if _addr(ytab) == 0 else _view(ytab, (int(cr[:, 1].max()) + H,), torch.int32).numpy()
        xt = None if _addr(xtab) == 0 else _view(xtab, (int(cr[:, 0].max()) + W,), torch.int32).numpy()
        m = None if _addr(mul) == 0 else _view(mul, (N, 1, H, W), torch.float32).numpy()
        out = IO.crop_flip_to_tensor(_view(src, (N, Hs, Ws, C), torch.uint8).numpy(), cr, H, W, mode, unknown_label, yt, xt, m)
        _view(dst, (N, C, H, W), torch.float32)[:] = torch.from_numpy(out)
        return 0

    def mg_onehot_labels(self, label, out, N, HW, nc, stream=None):
        from oracle import inputs_oracle as IO
        lab = _view(label, (N, 1, HW), torch.float32).numpy()
        _view(out, (N, nc, HW), torch.float32)[:] = torch.from_numpy(IO.onehot_labels(lab, nc))
        return 0

    def mg_orient_to_rgb_u8(self, orient, label, table, out, npix, stream=None):
        from oracle import inputs_oracle as IO
        o = _view(orient, (1, npix), torch.uint8).numpy()
        l = _view(label, (1, npix), torch.uint8).numpy()
        _view(out, (1, npix, 3), torch.uint8)[:] = torch.from_numpy(IO.trans_orient_to_rgb(o, l))
        return 0

    def mg_generate_hole_u8(self, mask, orient_mask, th, u, hole, info, N, H, W, stream=None):
        from oracle import inputs_oracle as IO
        mk = _view(mask, (N, H, W), torch.uint8).numpy()
        om = _view(orient_mask, (N, H, W), torch.uint8).numpy()
        thv, uv = _view(th, (N,), torch.float64), _view(u, (N,), torch.float64)
        out = _view(hole, (N, H, W), torch.uint8)
        inf = None if _addr(info) == 0 else _view(info, (N, 4), torch.int32)
        for n in range(N):
            nums = int((om[n] != 0).sum())
            ci = IO.hole_center_index(float(uv[n]), nums) if nums else 0
            out[n] = torch.from_numpy(IO.generate_hole(mk[n], om[n], float(thv[n]), ci))
            if inf is not None:
                if nums:
                    ys, xs = (om[n] != 0).nonzero()
                    inf[n] = torch.tensor([nums, int(ys[ci]), int(xs[ci]), int(int(float(thv[n]) * nums) / math.pi)], dtype=torch.int32)
                else:
                    inf[n] = torch.tensor([0, -1, -1, 0], dtype=torch.int32)
        return 0

    def mg_noise_octaves(self, fields, out, N, S, stream=None):
        from oracle import inputs_oracle as IO
        sizes = IO.noise_octave_sizes(S)
        per = sum(s * s * 3 for s in sizes)
        fv = _view(fields, (N, per), torch.float64).numpy()
        o = _view(out, (N, 3, S, S), torch.float32)
        for n in range(N):
            fl, off = [], 0
            for s in sizes:
                fl.append(fv[n, off:off + s * s * 3].reshape(s, s, 3))
                off += s * s * 3
            o[n] = torch.from_numpy(IO.generate_noise_from_fields(fl, S)).permute(2, 0, 1)
        return 0

    def mg_nearest_table(self, src, dst, table):
        from oracle import inputs_oracle as IO
        _view(table, (dst,), torch.int32)[:] = torch.from_numpy(IO.pil_nearest_table(src, dst))
        return 0

    # ---- per-pixel glue (mg_glue.hip): contracts stated with torch's own ops, the ones the host stack used to call ------------
    @staticmethod
    def _plane(ptr, n, nstride, H, W):
        """[N, H, W] view of planes that sit nstride floats apart (a channel slice of an NCHW tensor)."""
        base = _view(ptr, ((n - 1) * nstride + H * W,), torch.float32)
        return torch.as_strided(base, (n, H, W), (nstride, W, 1))

    def mg_nearest_pyramid(self, d, stream=None):
        import torch.nn.functional as F
        td = _TD[d.dtype]
        planes = torch.stack([self._plane(d.plane[c], d.N, d.nstride[c], d.H, d.W) for c in range(d.nplanes)], dim=1)   # [N, C, H, W]
        for lev in range(d.nlev):
            h, w = d.h[lev], d.w[lev]
            r = planes if (h, w) == (d.H, d.W) else F.interpolate(planes, size=(h, w), mode="nearest")
            o = _view(d.out[lev], (d.N, h, w, d.cout), td)
            o.zero_()
            o[..., :d.nplanes] = r.permute(0, 2, 3, 1).to(td)
        return 0

    def mg_pconv_mask(self, mask_in, N, H, W, k, s, p, scale, upd, stream=None):
        import torch.nn.functional as F
        m = _view(mask_in, (N, 1, H, W), torch.float32).double()
        ssum = F.conv2d(m, torch.ones(1, 1, k, k, dtype=torch.float64), stride=s, padding=p)       # partialconv2d.py:62-64
        ratio = (k * k) / (ssum + 1e-8)
        u = ssum.clamp(0, 1)
        h, w = ssum.shape[2], ssum.shape[3]
        _view(scale, (N, 1, h, w), torch.float32)[:] = (ratio * u).float()
        _view(upd, (N, 1, h, w), torch.float32)[:] = u.float()
        return 0

    def mg_pixel_affine(self, x, a, bias, b, dtype, P, C, y, stream=None):
        td = _TD[dtype]
        xv = _view(x, (P, C), td)
        av = _view(a, (P, 1), torch.float32).to(td)
        if _addr(bias):
            t = _view(bias, (1, C), torch.float32).to(td) * _view(b, (P, 1), torch.float32).to(td)     # rounded to the activation dtype
            out = (t.double() + xv.double() * av.double()).to(td)
        else:
            out = (xv.double() * av.double()).to(td)
        _view(y, (P, C), td)[:] = out
        return 0

    def mg_bg_compose(self, image, noise, hair, hair_nstride, dtype, N, H, W, k, mode, inp, back, stream=None):
        import torch.nn.functional as F
        td = _TD[dtype]
        hp = self._plane(hair, N, hair_nstride, H, W).unsqueeze(1)
        if mode == 0:
            bk = 1 - F.max_pool2d(hp, k, 1, k // 2)                                                  # encoder.py:288-297
        else:
            bk = hp.clone()
        img = _view(image, (N, 3, H, W), torch.float32) if _addr(image) else None
        noi = _view(noise, (N, 3, H, W), torch.float32) if _addr(noise) else None
        if noi is None:
            v = img * bk
        elif img is None:
            v = noi
        else:
            v = img * bk + noi * (1 - bk)
        o = _view(inp, (N, H, W, 8), td)
        o.zero_()
        o[..., :3] = v.permute(0, 2, 3, 1).to(td)
        _view(back, (N, H, W), torch.float32)[:] = bk[:, 0]
        return 0

    def mg_masked_mean_fill(self, x, w_in, w_out, w_norm, dtype, N, P, C, out, stream=None):
        xv = _view(x, (N, P, C), _TD[dtype]).double()
        wi, wo, wn = (_view(t, (N, P, 1), torch.float32).double() for t in (w_in, w_out, w_norm))
        area = wn.sum(dim=(1, 2)).clamp_min(1.0)
        mean = (xv * wi).sum(dim=1) / area[:, None]                                   # encoder.py:216-219
        _view(out, (N, P, C), torch.float32)[:] = (mean[:, None, :] * wo).float()
        return 0

    def _orient_terms(self, conf_raw, idx, label, label_ch, label_nstride, hair, hair_nstride, N, HW):
        cr = _view(conf_raw, (N, HW), torch.float32).double()
        ix = _view(idx, (N, HW), torch.uint8).double()
        hv = self._plane(hair, N, hair_nstride, 1, HW).reshape(N, HW).double()
        if label_ch == 2:
            base = _view(label, ((N - 1) * label_nstride + 2 * HW,), torch.float32)
            lab = torch.as_strided(base, (N, 2, HW), (label_nstride, HW, 1)).double()
            l0, l1 = lab[:, 0], lab[:, 1]
        else:
            a = self._plane(label, N, label_nstride, 1, HW).reshape(N, HW).double() / 255 * math.pi
            l0, l1 = torch.sin(2 * a), torch.cos(2 * a)
        conf = (torch.tanh(cr) + 1) / 2
        ang = ix * (math.pi / 32)
        f0, f1 = torch.sin(2 * ang), torch.cos(2 * ang)
        d0, d1 = f0 * conf * hv - l0 * hv, f1 * conf * hv - l1 * hv
        return cr, hv, conf, f0, f1, d0, d1

    def mg_orient_loss_fwd(self, conf_raw, idx, label, label_ch, label_nstride, hair, hair_nstride, N, HW, out, ws, stream=None):
        cr, hv, conf, f0, f1, d0, d1 = self._orient_terms(conf_raw, idx, label, label_ch, label_nstride, hair, hair_nstride, N, HW)
        o = _view(out, (3,), torch.float32)
        o[0] = float((d0.abs().sum() + d1.abs().sum()) / (2 * N * HW))                              # F.l1_loss over [N, 2, H, W]
        o[1] = float(-(torch.log(conf.clamp(0.001, 1)) * hv).sum() / hv.sum())
        o[2] = float(hv.sum())
        return 0

    def mg_orient_loss_bwd(self, conf_raw, idx, label, label_ch, label_nstride, hair, hair_nstride, g_orient, g_conf, fwd_out, N, HW, dconf, stream=None):
        cr, hv, conf, f0, f1, d0, d1 = self._orient_terms(conf_raw, idx, label, label_ch, label_nstride, hair, hair_nstride, N, HW)
        g0 = float(_view(g_orient, (1,), torch.float32)[0]) if _addr(g_orient) else 0.0
        g1 = float(_view(g_conf, (1,), torch.float32)[0]) if _addr(g_conf) else 0.0
        sh = float(_view(fwd_out, (3,), torch.float32)[2])
        dcf = g0 / (2 * N * HW) * (torch.sign(d0) * f0 + torch.sign(d1) * f1) * hv
        inside = ((conf >= 0.001) & (conf <= 1)).double()
        dcf = dcf - g1 / sh * hv / conf * inside
        t = torch.tanh(cr)
        _view(dconf, (N, HW), torch.float32)[:] = (dcf * (1 - t * t) * 0.5).float()
        return 0

    def mg_orient_rgb_table(self, table):
        from oracle import inputs_oracle as IO
        _view(table, (256, 3), torch.float64)[:] = torch.from_numpy(IO.orient_rgb_table())
        return 0

    def mg_noise_field_len(self, S):
        from oracle import inputs_oracle as IO
        return sum(s * s * 3 for s in IO.noise_octave_sizes(S))

    def mg_inputs_set_option(self, key, value):
        return 0

    def mg_bicubic_ksize(self, in_size, out_size):
        from oracle import inputs_oracle as IO
        return IO.pil_bicubic_table(in_size, out_size)[1].shape[1]

    def mg_bicubic_table(self, in_size, out_size, bounds, coef):
        from oracle import inputs_oracle as IO
        b, c = IO.pil_bicubic_table(in_size, out_size)
        _view(bounds, b.shape, torch.int32)[:] = torch.from_numpy(b)
        _view(coef, c.shape, torch.int32)[:] = torch.from_numpy(c)
        return 0

    def mg_resize_bicubic_u8(self, src, tmp, dst, xb, xc, kx, yb, yc, ky, N, Hs, Ws, Hd, Wd, C, stream=None):
        from oracle import inputs_oracle as IO
        s = _view(src, (N, Hs, Ws, C), torch.uint8).numpy()
        d = _view(dst, (N, Hd, Wd, C), torch.uint8)
        for n in range(N):
            d[n] = torch.from_numpy(IO.pil_bicubic_resize_u8(s[n], Hd, Wd))
        return 0
