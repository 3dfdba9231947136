"""BaseNetwork: the small contract every pluggable network honours
(reference: models/networks/base_network.py:10-59 -- option hook, parameter count, init)."""
from __future__ import annotations

import torch
import torch.nn as nn
from torch.nn import init


class BaseNetwork(nn.Module):
    #: dtype the HIP kernels compute and store activations in (float32 or bfloat16)
    compute_dtype = torch.float32

    @staticmethod
    def modify_commandline_options(parser, is_train):
        return parser

    # per-process caches living in the instance __dict__ (ctypes tables, weak references): never pickled / deep-copied
    _TRANSIENT = ("_mg_input_cache", "_mg_spectral_plan", "_mg_mask_chain")

    def __getstate__(self):
        state = dict(self.__dict__)
        for k in self._TRANSIENT:
            state.pop(k, None)
        return state

    def __deepcopy__(self, memo):
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k not in self._TRANSIENT:
                new.__dict__[k] = copy.deepcopy(v, memo)
        return new

    def set_compute_dtype(self, dtype):
        if dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("compute dtype must be torch.float32 or torch.bfloat16")
        for m in self.modules():
            if isinstance(m, BaseNetwork):
                m.compute_dtype = dtype
        return self

    def print_network(self):
        count = sum(p.numel() for p in self.parameters())
        print("Network [%s] was created. Total number of parameters: %.1f million. "
              "To see the architecture, do print(network)." % (type(self).__name__, count / 1e6))

    def init_weights(self, init_type="normal", gain=0.02):
        """Same policy as the reference: BatchNorm affine ~ N(1, gain); Conv*/Linear weights by
        `init_type`, biases zero.  Class-name matching is kept ('Conv' in the name) so that the
        HIP conv modules -- subclasses of nn.Conv2d named *Conv2d -- are initialised identically."""
        def visit(m):
            cls = m.__class__.__name__
            if "BatchNorm2d" in cls:
                if getattr(m, "weight", None) is not None:
                    init.normal_(m.weight.data, 1.0, gain)
                if getattr(m, "bias", None) is not None:
                    init.constant_(m.bias.data, 0.0)
            elif hasattr(m, "weight") and ("Conv" in cls or "Linear" in cls):
                if init_type == "normal":
                    init.normal_(m.weight.data, 0.0, gain)
                elif init_type == "xavier":
                    init.xavier_normal_(m.weight.data, gain=gain)
                elif init_type == "xavier_uniform":
                    init.xavier_uniform_(m.weight.data, gain=1.0)
                elif init_type == "kaiming":
                    init.kaiming_normal_(m.weight.data, a=0, mode="fan_in")
                elif init_type == "orthogonal":
                    init.orthogonal_(m.weight.data, gain=gain)
                elif init_type == "none":
                    m.reset_parameters()
                else:
                    raise NotImplementedError("initialization method [%s] is not implemented" % init_type)
                if getattr(m, "bias", None) is not None:
                    init.constant_(m.bias.data, 0.0)

        self.apply(visit)
        for child in self.children():
            if hasattr(child, "init_weights"):
                child.init_weights(init_type, gain)
