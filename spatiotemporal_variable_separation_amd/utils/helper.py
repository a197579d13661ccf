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
