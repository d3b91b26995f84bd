"""setup_dsdf -- loader mirroring the reference sdfrenderer/deepsdf/workspace.py:167-195 (only the loader is on the path)."""
import importlib
import json
import os

import torch


def setup_dsdf(dir, mode='eval', precision=torch.float32):
    """Load `<x>.json` specs + `<x>.pt` state (keys may carry the DataParallel 'module.' prefix, workspace.py:176-180).

    Returns (decoder, latent_size).  The HIP path computes in float32; `precision` other than float32 is rejected rather
    than silently changing the arithmetic (the reference default float16 path is SURVEY.md §8 config 5, not built yet).
    """
    specs_filename = os.path.splitext(dir)[0] + '.json'
    if not os.path.isfile(specs_filename):
        raise Exception('The experiment directory does not include specifications file "specs.json"')
    specs = json.load(open(specs_filename))
    arch = importlib.import_module("sdflabel_amd.deepsdf.networks." + specs["NetworkArch"])
    latent_size = specs["CodeLength"]
    net_specs = dict(specs["NetworkSpecs"])
    net_specs.pop('samples_per_scene', None)
    decoder = arch.Decoder(latent_size, **net_specs)
    saved = torch.load(dir, map_location="cpu")
    state = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in saved["model_state_dict"].items()}
    decoder.load_state_dict(state)
    if precision != torch.float32:
        raise NotImplementedError("sdflabel_amd computes the decoder in float32 (requested %s)" % precision)
    decoder.to(dtype=torch.float32)
    if mode == 'train':
        decoder.train()
    else:
        decoder.eval()
    return decoder, latent_size
