"""CPU: the whole host stack (modules, autograd Functions, weight packing, losses) on the C-ABI
contract emulator, checked against the fixtures the reference produced and against the
reference's state_dict contract."""
import json
import os

import pytest
import torch

import parity_utils as PU


def test_state_dict_contract_matches_reference():
    """Checkpoint compatibility: same keys, same shapes as the reference at the BASELINE width."""
    from michigan_amd import networks
    from michigan_amd.model import default_options
    with open(os.path.join(PU.GOLDEN, "state_dict_contract.json")) as fh:
        contract = json.load(fh)
    opt = default_options(gpu_ids=[])
    with torch.device("meta"):
        nets = {"G": networks.SPADEBGenerator(opt), "D": networks.MultiscaleDiscriminator(opt), "VGG": networks.VGG19()}
    for name, net in nets.items():
        mine = {k: list(v.shape) for k, v in net.state_dict().items()}
        assert list(mine.keys()) == list(contract[name].keys()), name
        assert mine == contract[name], name


def test_name_lookup_and_option_hooks():
    import argparse
    from michigan_amd import networks
    assert networks.find_network_using_name("spadeb", "generator") is networks.SPADEBGenerator
    assert networks.find_network_using_name("multiscale", "discriminator") is networks.MultiscaleDiscriminator
    with pytest.raises(ValueError):
        networks.find_network_using_name("nonexistent", "generator")
    p = argparse.ArgumentParser()
    p.add_argument("--norm_G", default="spectralinstance")
    networks.SPADEBGenerator.modify_commandline_options(p, True)
    assert p.parse_args([]).norm_G == "spectralspadesyncbatch3x3"


def test_product_has_no_cpu_path():
    """Without the emulator installed, a CPU call must fail loudly rather than fall back."""
    from michigan_amd import _cabi, ops
    prev = _cabi.set_backend(None)
    try:
        x = torch.randn(1, 4, 4, 8)
        w = torch.randn(8, 8, 3, 3)
        with pytest.raises(Exception):
            ops.conv2d(x, w, None, padding=1)            # real library + host pointer -> HIP error, not a result
    finally:
        _cabi.set_backend(prev)


def test_generator_matches_reference_golden(emulator_backend):
    res = PU.run_generator("cpu")
    PU.compare(res, PU.golden("generator_ngf16_c128.npz"), atol_out=5e-5, rtol_stat=1e-4, rtol_grad=2e-3)


def test_discriminator_vgg_losses_match_reference_golden(emulator_backend):
    res = PU.run_discriminator_vgg("cpu")
    PU.compare(res, PU.golden("discriminator_vgg_ngf16_c128.npz"), atol_out=5e-5, rtol_stat=1e-4, rtol_grad=2e-3)


def test_inference_config0_matches_reference_golden(emulator_backend):
    """BASELINE configs[0] (inference.py: eval mode, --add_feat_zeros zero-padded canvas, fixed mask dilation)."""
    import numpy as np
    res, gold = PU.run_inference_config0("cpu"), PU.golden("inference_ngf16_c64.npz")
    assert res["out_padded"].shape == gold["out_padded"].shape
    assert np.abs(res["out_padded"] - gold["out_padded"]).max() < 5e-5
    assert np.abs(res["out"] - gold["out"]).max() < 5e-5
