"""Decoder loader -- mirror of reference deep_sdf/workspace.py:202-223 (config_decoder).

Reads the reference's on-disk format unchanged: <dir>/specs.json + <dir>/ModelParameters/<checkpoint>.pth
whose "model_state_dict" carries DataParallel's `module.` prefix.  Instead of building an nn.Module it
folds weight-norm and returns a deep_sdf.deep_sdf_decoder.Decoder bound to the HIP engine.
"""
import json
import os

model_params_subdir = "ModelParameters"
specifications_filename = "specs.json"


def config_decoder(experiment_directory, checkpoint="latest"):
    specs_filename = os.path.join(experiment_directory, specifications_filename)
    if not os.path.isfile(specs_filename):
        raise Exception('The experiment directory does not include specifications file "specs.json"')
    with open(specs_filename) as f:
        specs = json.load(f)
    if specs["NetworkArch"] != "deep_sdf_decoder":
        raise NotImplementedError("NetworkArch %r" % specs["NetworkArch"])
    from .deep_sdf_decoder import Decoder
    import torch  # container format of the checkpoint only
    path = os.path.join(experiment_directory, model_params_subdir, checkpoint + ".pth")
    saved = torch.load(path, map_location="cpu")
    decoder = Decoder(specs["CodeLength"], **specs["NetworkSpecs"])
    decoder.load_state_dict(saved["model_state_dict"])
    return decoder.cuda().eval()


def decoder_from_state_dict(state_dict, specs, device=None):
    """Convenience for tests / bench: build the decoder from an in-memory state dict (numpy or torch)."""
    from .deep_sdf_decoder import Decoder
    return Decoder(specs["CodeLength"], state_dict=state_dict, device=device, **specs["NetworkSpecs"])
