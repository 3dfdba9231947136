"""michigan_amd: the MichiGAN SPADEB training hot path on MI355X (hand-written HIP behind a C ABI).

Importing the package sets the one HIP-runtime knob the step depends on.  The runtime copies every launch's
kernel arguments into a ring ("kernarg pool"); when the ring is full the *host* blocks until the GPU has drained
it.  With the default pool a G+D step (about 4000 launches, many carrying 300-900 byte argument blocks) fills it
half-way through the backward pass, so the host can never enqueue ahead of the GPU and every launch-bound stretch
shows up as GPU idle time (tools/queue_depth_probe.py, tools/cpu_floor.py measure this).  64 MiB removes the stall.
The variable is read when the HIP runtime initialises, i.e. at the first device call of the process, so the package
has to be imported before that; a value already present in the environment is respected.
"""
import os as _os

_os.environ.setdefault("HSA_KERNARG_POOL_SIZE", str(64 << 20))
