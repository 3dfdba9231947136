"""How does the bs-4 step (the 8-GPU target's per-GPU batch) react to a contended host?  VERDICT r5 "next" 6(b).
One real rank drives the GPU; beside it N twin processes burn a host core each the way a rank's enqueue thread does (a pure-Python loop:
interpreter-bound, GIL held, no GPU).  On an 8-GPU node the seven other ranks are exactly such neighbours (plus RCCL proxy threads).
Prints ms/step for N = 0, 7, 15 and the host's core count / affinity.       tools/host_contention.py [bs]"""
import multiprocessing as mp
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def burner(stop):
    x = 0
    while not stop.is_set():
        for i in range(20000):
            x = (x * 1103515245 + i) & 0xFFFFFFFF


def main():
    import michigan_amd  # noqa: F401
    import torch
    from michigan_amd.model import Pix2PixTrainer, default_options
    from michigan_amd.synth import synth_batch
    bs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    tr = Pix2PixTrainer(default_options(crop_size=512, gpu_ids=[0], compute_dtype="bf16"))
    data = {k: v.cuda() for k, v in synth_batch(bs, 512, seed=1234).items()}

    def step():
        tr.run_generator_one_step(data); tr.run_discriminator_one_step(data)

    def timed(k=10):
        for _ in range(3):
            step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(k):
            step()
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        return 1e3 * (t1 - t0) / k, 1e3 * (t2 - t0) / k
    print("host: os.cpu_count() = %s, affinity = %d cores; bs %d" % (os.cpu_count(), len(os.sched_getaffinity(0)), bs))
    ctx = mp.get_context("spawn")
    for n in (0, 7, 15, 0):
        stop = ctx.Event()
        ps = [ctx.Process(target=burner, args=(stop,), daemon=True) for _ in range(n)]
        for p in ps:
            p.start()
        time.sleep(1.0 if n else 0.0)
        enq, tot = timed()
        stop.set()
        for p in ps:
            p.join(10)
        print("%2d burner process(es) beside the rank: host enqueue %.2f ms/step, step %.2f ms" % (n, enq, tot))


if __name__ == "__main__":
    main()
