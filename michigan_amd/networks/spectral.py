"""torch.nn.utils.spectral_norm with the weight computation on the fused HIP path.

`spectral_norm(module)` is torch's own call -- same parameters (`weight_orig`), buffers (`weight_u`, `weight_v`),
state-dict hooks and in-place power-iteration semantics (torch/nn/utils/spectral_norm.py) -- after which the hook
object's class is swapped for a subclass whose `compute_weight` runs ops.spectral_weight for the common case
(dim 0, one power iteration, fp32 weights with a multiple-of-4 element count); everything else falls through to torch.
Reference call sites: architecture.py:39-42, normalization.py:28-29.
"""
from __future__ import annotations

import torch
from torch.nn.utils import spectral_norm as _torch_spectral_norm
from torch.nn.utils.spectral_norm import SpectralNorm

from .. import _cabi as C
from .. import ops


class FusedSpectralNorm(SpectralNorm):
    _mg_pre = None           # (do_power_iteration, W_sn) prepared for the coming forward by SpectralPlan.prepare

    def fusable(self, weight) -> bool:
        return (self.dim == 0 and self.n_power_iterations == 1 and weight.dtype == torch.float32 and weight.numel() % 4 == 0
                and weight.is_contiguous())

    def compute_weight(self, module, do_power_iteration):
        weight = getattr(module, self.name + "_orig")
        pre, self._mg_pre = self._mg_pre, None
        if pre is not None and pre[0] == bool(do_power_iteration):
            return pre[1]
        if not self.fusable(weight):
            return super().compute_weight(module, do_power_iteration)
        return ops.spectral_weight(weight, getattr(module, self.name + "_u"), getattr(module, self.name + "_v"),
                                   do_power_iteration, self.eps)


def spectral_norm(module, name="weight", n_power_iterations=1, eps=1e-12, dim=None):
    module = _torch_spectral_norm(module, name=name, n_power_iterations=n_power_iterations, eps=eps, dim=dim)
    for hook in module._forward_pre_hooks.values():
        if type(hook) is SpectralNorm and hook.name == name:
            hook.__class__ = FusedSpectralNorm
    return module


BATCHED = True           # SpectralPlan (False: per-layer compute_weight, the path stand-alone modules and a network's first pass take; tests)


class SpectralPlan:
    """All spectral-normed convolutions of one network, prepared together at the top of its forward:
    mg_sn_power_iteration (4 launches for every layer's u / v / sigma) + mg_pack_weights (1 launch: every layer's W / sigma
    and the GEMM images its convolution will ask for) instead of ~7 launches per layer.  The per-module forward pre-hooks
    then just pick up what was prepared.  The first iteration runs the per-layer path, which records the image geometries."""

    def __init__(self, root):
        self.entries = []
        for m in root.modules():
            for hook in getattr(m, "_forward_pre_hooks", {}).values():
                if isinstance(hook, FusedSpectralNorm) and hook.name == "weight":
                    self.entries.append((m, hook))
        self._sig = None
        self._static = None

    def _signature(self):
        return tuple((getattr(m, "weight_orig").data_ptr(), m.weight_u.data_ptr(), m.weight_v.data_ptr(),
                      tuple(sorted(getattr(getattr(m, "weight_orig"), "_mg_pack_geom", ()), key=repr))) for m, _ in self.entries)

    def _build(self, dev):
        be = C.backend()
        n = len(self.entries)
        layers = ops.DeviceTable(C.SnLayer, n, dev)
        k1, k3, scratch, off = [], [], [], 0
        for m, _ in self.entries:
            w = m.weight_orig
            rows, cols = w.shape[0], w.numel() // w.shape[0]
            k1.append(int(be.mg_sn_layer_blocks(rows, cols, 0)))
            k3.append(int(be.mg_sn_layer_blocks(rows, cols, 1)))
            chunks = int(be.mg_sn_layer_blocks(rows, cols, 2))
            c4, r4 = (cols + 3) // 4 * 4, (rows + 3) // 4 * 4                       # every vector starts on a 16-byte boundary
            scratch.append((off, off + c4, off + c4 + r4))                          # t1, t2, partial
            off += c4 + r4 + (chunks * cols + 3) // 4 * 4
        self._scratch = torch.empty(off, dtype=torch.float32, device=dev)
        # per-forward buffer layout (floats): [sigma | u_copy | v_copy] per layer, then W_sn, then the images (bytes handled below)
        self._static = dict(layers=layers, k1=k1, k3=k3, scratch=scratch, map1=ops.block_map(k1, dev), map3=ops.block_map(k3, dev))
        jobs = []                                                                   # (layer index, mode 2 | geometry)
        for i, (m, _) in enumerate(self.entries):
            jobs.append((i, None))
            for g in sorted(getattr(m.weight_orig, "_mg_pack_geom", ()), key=repr):
                jobs.append((i, g))
        self._static["jobs"] = jobs

    def prepare(self, root) -> bool:
        if not BATCHED or not self.entries:
            return False
        ws = [m.weight_orig for m, _ in self.entries]
        if any(not h.fusable(w) or w.shape[2] * w.shape[3] > 49 or (not w.is_cuda and C.backend().name == "hip")
               for (m, h), w in zip(self.entries, ws)):
            return False
        if any(not getattr(w, "_mg_pack_geom", None) for w in ws):
            return False                                     # first pass: the per-layer path records what the convolutions need
        training = [m.training for m, _ in self.entries]
        if any(t != training[0] for t in training):
            return False
        grad = torch.is_grad_enabled()
        if grad and not all(any(g[3] == 1 for g in w._mg_pack_geom) for w in ws):
            return False                                     # data-gradient images not seen yet (no backward has run so far)
        dev = ws[0].device
        sig = self._signature()
        if sig != self._sig:
            self._build(dev)
            self._sig = sig
        st, be, n = self._static, C.backend(), len(self.entries)
        power = bool(training[0])
        # ---- this forward's outputs: one fp32 buffer (sigma, private u / v copies, W_sn) + one byte buffer (GEMM images) ----------------
        f_off, fl = [], 0
        for w in ws:
            rows, cols = w.shape[0], w.numel() // w.shape[0]
            f_off.append((fl, fl + 4, fl + 4 + rows, fl + 4 + rows + cols))         # sigma (padded to 4), u_copy, v_copy, W_sn
            fl += 4 + rows + cols + w.numel()
            fl = (fl + 3) // 4 * 4
        fbuf = torch.empty(fl, dtype=torch.float32, device=dev)
        jobs = [(i, g) for (i, g) in st["jobs"] if g is None or g[3] == 0 or grad]
        img_off, nb = [], 0
        for i, g in jobs:
            if g is None:
                img_off.append(None)
                continue
            dtype, rows_p, cols_p, mode = g
            w = ws[i]
            nbytes = w.shape[2] * w.shape[3] * rows_p * cols_p * (2 if dtype == torch.bfloat16 else 4)
            img_off.append(nb)
            nb += (nbytes + 255) // 256 * 256
        ibuf = torch.empty(max(nb, 256), dtype=torch.uint8, device=dev)
        fp, ip, sp = fbuf.data_ptr(), ibuf.data_ptr(), self._scratch.data_ptr()
        rows_tab = st["layers"].begin_update()
        b1 = b3 = 0
        for i, ((m, _), w) in enumerate(zip(self.entries, ws)):
            r = rows_tab[i]
            o_sig, o_u, o_v, o_w = f_off[i]
            t1, t2, part = st["scratch"][i]
            r.w, r.u, r.v = w.data_ptr(), m.weight_u.data_ptr(), m.weight_v.data_ptr()
            r.u_copy, r.v_copy = (fp + 4 * o_u, fp + 4 * o_v) if (grad and power) else (None, None)
            r.sigma, r.t1, r.t2, r.partial = fp + 4 * o_sig, sp + 4 * t1, sp + 4 * t2, sp + 4 * part
            r.rows, r.cols, r.first_block_k1, r.first_block_k3 = w.shape[0], w.numel() // w.shape[0], b1, b3
            b1 += st["k1"][i]
            b3 += st["k3"][i]
        stream = ops._stream(fbuf)
        be.mg_sn_power_iteration(st["layers"].upload(), n, ops._p(st["map1"]), b1, ops._p(st["map3"]), b3, 1 if power else 0,
                                 float(self.entries[0][1].eps), stream)
        key = (grad,)
        jt = st.get(("jobtable",) + key)
        if jt is None:
            counts = []
            for (i, g) in jobs:
                w = ws[i]
                taps = w.shape[2] * w.shape[3]
                counts.append(int(be.mg_pack_job_blocks(2, w.shape[0], w.shape[1], taps, 0, 0) if g is None
                                  else be.mg_pack_job_blocks(g[3], w.shape[0], w.shape[1], taps, g[1], g[2])))
            jt = st[("jobtable",) + key] = (ops.DeviceTable(C.PackJob, len(jobs), dev), counts, ops.block_map(counts, dev))
        table, counts, bmap = jt
        first = 0
        for row, (i, g), off, cnt in zip(table.begin_update(), jobs, img_off, counts):
            w = ws[i]
            row.w0, row.w1, row.sigma = w.data_ptr(), None, fp + 4 * f_off[i][0]
            row.cout, row.cin, row.taps = w.shape[0], w.shape[1], w.shape[2] * w.shape[3]
            if g is None:
                row.dst, row.mode, row.dtype, row.rows_p, row.cols_p = fp + 4 * f_off[i][3], 2, C.MG_F32, 0, 0
            else:
                dtype, rows_p, cols_p, mode = g
                row.dst, row.mode, row.rows_p, row.cols_p = ip + off, mode, rows_p, cols_p
                row.dtype = C.MG_BF16 if dtype == torch.bfloat16 else C.MG_F32
            row.first_block = first
            first += cnt
        be.mg_pack_weights(table.upload(), len(jobs), ops._p(bmap), first, stream)
        # ---- hand the results to the hooks ------------------------------------------------------------------------------------
        images = [dict() for _ in ws]
        for (i, g), off in zip(jobs, img_off):
            if g is not None:
                dtype, rows_p, cols_p, mode = g
                w = ws[i]
                taps = w.shape[2] * w.shape[3]
                nel = taps * rows_p * cols_p
                images[i][g] = ibuf[off:off + nel * (2 if dtype == torch.bfloat16 else 4)].view(dtype).view(taps, rows_p, cols_p)
        for i, ((m, hook), w) in enumerate(zip(self.entries, ws)):
            o_sig, o_u, o_v, o_w = f_off[i]
            rows, cols = w.shape[0], w.numel() // w.shape[0]
            out = fbuf[o_w:o_w + w.numel()].view(w.shape)
            sigma = fbuf[o_sig:o_sig + 1]
            if grad and w.requires_grad:
                if power:
                    uc, vc = fbuf[o_u:o_u + rows], fbuf[o_v:o_v + cols]
                else:
                    uc, vc = m.weight_u.clone(), m.weight_v.clone()
                w_sn = ops._SpectralPrecomputedFn.apply(w, uc, vc, sigma, [out])
                w_sn._mg_sn = (w, out, uc, vc, sigma)
            else:
                w_sn = out
            w_sn._mg_sn_origin = w
            w_sn._mg_packed = images[i]
            hook._mg_pre = (power, w_sn)
        return True


def prepare(root) -> None:
    """Call at the top of a network's forward: batch this pass's spectral-norm work (no-op when nothing is batchable)."""
    plan = root.__dict__.get("_mg_spectral_plan")
    if plan is None:
        plan = root.__dict__["_mg_spectral_plan"] = SpectralPlan(root)
    plan.prepare(root)
