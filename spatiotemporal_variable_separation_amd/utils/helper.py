"""Checkpoint and config helpers (reference: utils/helper.py:22-78): same four file names, whole-module pickles."""
import json
import os

import torch
import yaml


def save(elem_xp_path, sep_net, epoch_number=None):
    """Write ov_Et/ov_Es/decoder/t_resnet `.pt` files (helper.py:22-33).  Unlike the reference this does not retry
    forever on failure: an I/O error is raised to the caller."""
    append = f'_{epoch_number}' if epoch_number is not None else ''
    os.makedirs(elem_xp_path, exist_ok=True)
    for stem, module in (('ov_Et', sep_net.Et), ('ov_Es', sep_net.Es), ('decoder', sep_net.decoder),
                         ('t_resnet', sep_net.t_resnet)):
        torch.save(module, os.path.join(elem_xp_path, f'{stem}{append}.pt'))


class DotDict(dict):
    """Dictionary with attribute access; missing keys read as None (helper.py:54-60)."""
    __getattr__ = dict.get
    __setattr__ = dict.__setitem__
    __delattr__ = dict.__delitem__


def load_yaml(path):
    with open(path, 'r') as f:
        return DotDict(yaml.safe_load(f))


def load_json(path):
    with open(path, 'r') as f:
        return DotDict(json.load(f))


def load_model(xp_dir, sep_net, epoch_number=None, map_location='cpu'):
    """Load the four checkpoint files written by `save` -- or by the REFERENCE's `save` (utils/helper.py:22-33, whole-module
    pickles of `var_sep.networks.*` classes; unpickling those needs the reference package importable) -- into `sep_net`.

    Only `state_dict()`s cross over: keys and shapes are identical by construction, so reference-trained weights run on
    the HIP path unchanged.  torch >= 2.6 defaults to weights_only=True, which rejects
    whole-module pickles; they are loaded with weights_only=False like the reference's test/utils.py:10-13 intends."""
    append = f'_{epoch_number}' if epoch_number is not None else ''
    for stem, module in (('ov_Et', sep_net.Et), ('ov_Es', sep_net.Es), ('decoder', sep_net.decoder),
                         ('t_resnet', sep_net.t_resnet)):
        obj = torch.load(os.path.join(xp_dir, f'{stem}{append}.pt'), map_location=map_location, weights_only=False)
        state = obj.state_dict() if hasattr(obj, 'state_dict') else obj
        module.load_state_dict(state, strict=True)
    return sep_net
