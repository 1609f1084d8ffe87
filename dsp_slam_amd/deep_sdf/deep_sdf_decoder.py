"""DeepSDF decoder object for the MI355X path.

Mirrors the role of reference deep_sdf/deep_sdf_decoder.py:9-110 (class Decoder): it is the `decoder`
object that C++ creates once (src/System.cc:97) and hands to Optimizer / MeshExtractor.  Here it
holds the weight-norm-folded layers and (lazily) the HIP Engine that owns the packed weights on the
GPU.  Calling it evaluates the network on the GPU -- there is no PyTorch module underneath.
"""
import os

import numpy as np

from dsp_slam_amd import engine as _engine


def fold_weight_norm(state_dict, n_linear):
    """W = g * v / ||v||_row for weight-normed layers (nn.utils.weight_norm, deep_sdf_decoder.py:49-54)."""
    sd = {}
    for k, v in state_dict.items():
        a = v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
        sd[k[7:] if k.startswith("module.") else k] = np.asarray(a, np.float32)   # DataParallel prefix, workspace.py:214-218
    layers = []
    for k in range(n_linear):
        name = "lin%d" % k
        if name + ".weight_v" in sd:
            v = sd[name + ".weight_v"]
            g = sd[name + ".weight_g"].reshape(-1, 1)
            nrm = np.sqrt((v.astype(np.float64) ** 2).sum(axis=1, keepdims=True)).astype(np.float32)
            w = (v * (g / nrm)).astype(np.float32)
        else:
            w = sd[name + ".weight"]
        layers.append((np.ascontiguousarray(w, np.float32), np.ascontiguousarray(sd[name + ".bias"], np.float32)))
    return layers


class Decoder(object):
    def __init__(self, latent_size, dims, dropout=None, dropout_prob=0.0, norm_layers=(), latent_in=(), weight_norm=False,
                 xyz_in_all=None, use_tanh=False, latent_dropout=False, state_dict=None, device=None):
        if xyz_in_all or use_tanh or latent_dropout:
            raise NotImplementedError("decoder variant not supported by the MI355X path (xyz_in_all/use_tanh/latent_dropout)")
        if not weight_norm and norm_layers:
            raise NotImplementedError("LayerNorm decoders (weight_norm=False) are not supported by the MI355X path")
        self.latent_size = int(latent_size)
        self.dims = list(dims)
        self.latent_in = tuple(latent_in)
        self.num_layers = len(dims) + 2
        self.layers = None
        self._engine = None
        self._device = device
        if state_dict is not None:
            self.load_state_dict(state_dict)

    def load_state_dict(self, state_dict):
        self.layers = fold_weight_norm(state_dict, len(self.dims) + 1)
        self._engine = None

    # nn.Module look-alikes the reference loader calls (deep_sdf/workspace.py:219-221)
    def cuda(self, device=None):
        if device is not None:
            self._device = device
        return self

    def eval(self):
        return self

    @property
    def engine(self):
        if self._engine is None:
            if self.layers is None:
                raise RuntimeError("decoder has no weights loaded")
            dev = self._device
            if dev is None:
                dev = int(os.environ.get("DSP_DEVICE", os.environ.get("LOCAL_RANK", "0")))
            self._engine = _engine.Engine(self.layers, self.latent_in, self.latent_size, device=int(dev))
        return self._engine

    def __call__(self, inputs):
        """(..., latent+3) -> (..., 1): rows are grouped by latent code, each group decoded on the GPU."""
        import torch
        x = inputs.detach().cpu().numpy() if hasattr(inputs, "detach") else np.asarray(inputs)
        shape = x.shape[:-1]
        x = np.ascontiguousarray(x.reshape(-1, x.shape[-1]), np.float32)
        out = np.zeros(x.shape[0], np.float32)
        codes, inv = np.unique(x[:, :-3], axis=0, return_inverse=True)
        for i, c in enumerate(codes):
            sel = np.where(inv.reshape(-1) == i)[0]
            out[sel] = self.engine.decode_sdf(c, x[sel, -3:])
        return torch.from_numpy(out.reshape(shape + (1,)))

    forward = __call__
