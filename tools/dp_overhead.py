"""Where does the data-parallel overhead go?  One rank over RCCL (MG_DP_FORCE=1), collectives toggled in-process.
Run: MG_DP_FORCE=1 python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 tools/dp_overhead.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import michigan_amd  # noqa: F401
import torch, torch.distributed as dist
from michigan_amd import ops
from michigan_amd.model import Pix2PixTrainer, default_options
from michigan_amd.synth import synth_batch
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
opt = default_options(crop_size=512, gpu_ids=[0], compute_dtype="bf16")
tr = Pix2PixTrainer(opt)
data = {k: v.cuda() for k, v in synth_batch(8, 512, seed=1234).items()}
grp = ops.SYNC_BN_GROUP
def step():
    tr.run_generator_one_step(data); tr.run_discriminator_one_step(data)
def timed(tag):
    for _ in range(2): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(6): step()
    torch.cuda.synchronize(); print(f"{tag}: {(time.perf_counter() - t0) / 6 * 1e3:.2f} ms/step", flush=True)
for rep in range(2):
    ops.SYNC_BN_GROUP = grp; tr.optimizer_G.dp = tr.optimizer_D.dp = True
    timed("sync-BN reductions + gradient buckets")
    ops.SYNC_BN_GROUP = None
    timed("gradient buckets only               ")
    ops.SYNC_BN_GROUP = grp; tr.optimizer_G.dp = tr.optimizer_D.dp = False
    timed("sync-BN reductions only             ")
    ops.SYNC_BN_GROUP = None
    timed("no collectives                      ")
dist.destroy_process_group()
