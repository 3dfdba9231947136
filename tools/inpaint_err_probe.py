"""Which op moves the bf16 error of the frozen in-painting net against the reference's fixture (tests/golden/inpaint_c64.npz)?  (GPU box)
mean |out - golden| of the bf16 module with: the flash attention kernel vs softmax(q k^T) v in fp32 torch ops on the same bf16 q / k / v
(the round-3 path), and the skip connection added inside the instance-norm apply launch vs as a separate bf16 add (round 3)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import michigan_amd  # noqa: F401
from michigan_amd import ops
import test_inpaint as TI

cfg, x, _, _ = TI._inputs()
gold = torch.from_numpy(np.load(os.path.join(TI.PU.GOLDEN, "inpaint_c64.npz"))["out"])
real_attn, real_in = ops.self_attention, ops.instance_norm_act_infer


def torch_attn(q, k, v, out=None):
    o = torch.bmm(torch.softmax(torch.bmm(q.float(), k.float().transpose(1, 2)), dim=-1), v.float()).to(v.dtype)
    if out is None:
        return o
    out.copy_(o)
    return out


def split_in(x_, *, eps=1e-5, act=ops.ACT_NONE, slope=0.2, resid=None):
    y = real_in(x_, eps=eps, act=act, slope=slope)
    return y if resid is None else resid + y


for dt in (torch.bfloat16, torch.float32):
    for an, af in (("flash kernel", real_attn), ("torch fp32 bmm/softmax", torch_attn)):
        for rn, rf in (("skip fused in the apply", real_in), ("separate add", split_in)):
            ops.self_attention, ops.instance_norm_act_infer = af, rf
            ig, _ = TI._module("cuda", dt)
            with torch.no_grad():
                err = (ig(x.cuda()).cpu() - gold).abs()
            print("%-8s attention: %-24s residual: %-24s mean |err| %.5f  max %.4f" % (str(dt).split(".")[1], an, rn, err.mean().item(), err.max().item()), flush=True)
ops.self_attention, ops.instance_norm_act_infer = real_attn, real_in
