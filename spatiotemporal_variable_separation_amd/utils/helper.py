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


class _ReferenceClassStub(torch.nn.Module):
    """Stand-in for a class of the reference package that has no namesake here: the pickled `__dict__` (`_modules`, `_parameters`,
    `_buffers`) is all `state_dict()` needs."""

    def forward(self, *args, **kwargs):
        raise RuntimeError('this module was restored from a reference checkpoint only to read its state_dict')


def _product_classes():
    """Class name -> this package's class, for every network class a reference checkpoint can hold (reference `var_sep/networks/*.py`:
    conv.py:63-564, mlp.py:44, mlp_encdec.py:25-50, resnet.py:22-88, utils.py:21, model.py:20).  The ConvRes* classes live in
    networks/conv.py here and in resnet.py there, so the lookup goes by class name, not by module path."""
    from ..networks import conv, mlp, mlp_encdec, model, resnet, utils
    table = {}
    for mod in (utils, mlp, mlp_encdec, resnet, conv, model):
        for name, obj in vars(mod).items():
            if isinstance(obj, type) and issubclass(obj, torch.nn.Module) and obj.__module__ == mod.__name__:
                table[name] = obj
    return table


class _ReferencePickle:
    """`pickle_module` for torch.load: unpickles whole-module checkpoints written by the REFERENCE's `save` (utils/helper.py:22-33:
    `torch.save(sep_net.Et, ...)`, i.e. pickles that name classes `var_sep.networks.*`) WITHOUT the reference package being importable:
    `find_class` maps every `var_sep.*` class onto this package's class of the same name (same attribute and sub-module names, so the
    restored object is a working module of the HIP path) or onto a bare nn.Module stand-in.  Everything else resolves as usual."""
    import pickle as _pickle
    __name__ = 'pickle'
    load, loads, dump, dumps = _pickle.load, _pickle.loads, _pickle.dump, _pickle.dumps
    HIGHEST_PROTOCOL, DEFAULT_PROTOCOL = _pickle.HIGHEST_PROTOCOL, _pickle.DEFAULT_PROTOCOL
    Pickler, PickleError, UnpicklingError = _pickle.Pickler, _pickle.PickleError, _pickle.UnpicklingError

    class Unpickler(_pickle.Unpickler):
        _table = None

        def find_class(self, module, name):
            if module == 'var_sep' or module.startswith('var_sep.'):
                cls = type(self)
                if cls._table is None:
                    cls._table = _product_classes()
                hit = cls._table.get(name)
                if hit is not None:
                    return hit
                return type(name, (_ReferenceClassStub,), {'__module__': __name__, '_reference_path': module + '.' + name})
            return super().find_class(module, name)


def load_module_file(path, map_location='cpu'):
    """One checkpoint file -> the object it holds: a state dict, a module of this package, or a whole-module pickle of the reference
    (its classes mapped onto this package's, see _ReferencePickle).  torch >= 2.6 defaults to weights_only=True, which rejects
    whole-module pickles; they are loaded with weights_only=False like the reference's test/utils.py:10-13 intends."""
    return torch.load(path, map_location=map_location, weights_only=False, pickle_module=_ReferencePickle)


def load_model(xp_dir, sep_net, epoch_number=None, map_location='cpu'):
    """Load the four checkpoint files written by `save` -- or by the REFERENCE's `save` (utils/helper.py:22-33, whole-module
    pickles of `var_sep.networks.*` classes) -- into `sep_net`.  The reference package does NOT have to be importable: its class
    paths are resolved onto this package's classes while unpickling (`_ReferencePickle`).

    Only `state_dict()`s cross over: keys and shapes are identical by construction, so reference-trained weights run on
    the HIP path unchanged."""
    append = f'_{epoch_number}' if epoch_number is not None else ''
    for stem, module in (('ov_Et', sep_net.Et), ('ov_Es', sep_net.Es), ('decoder', sep_net.decoder),
                         ('t_resnet', sep_net.t_resnet)):
        obj = load_module_file(os.path.join(xp_dir, f'{stem}{append}.pt'), map_location=map_location)
        state = obj.state_dict() if hasattr(obj, 'state_dict') else obj
        module.load_state_dict(state, strict=True)
    return sep_net


def load_sep_net(xp_dir, nt_cond, skipco, epoch_number=None, map_location='cpu'):
    """The reference's `test/utils.py:8-16 load_model(args)`: build a SeparableNetwork straight from the four files (no architecture
    flags needed: the pickles carry the module trees), in eval mode.  Works for checkpoints of this package and of the reference."""
    from ..networks.model import SeparableNetwork
    append = f'_{epoch_number}' if epoch_number is not None else ''
    mods = {stem: load_module_file(os.path.join(xp_dir, f'{stem}{append}.pt'), map_location=map_location)
            for stem in ('ov_Es', 'ov_Et', 't_resnet', 'decoder')}
    for stem, m in mods.items():
        if isinstance(m, _ReferenceClassStub) or not isinstance(m, torch.nn.Module):
            raise TypeError(f'{stem}: {type(m).__name__} is not a network class of this package; use load_model() with a network built '
                            'from the architecture flags')
    sep_net = SeparableNetwork(mods['ov_Es'], mods['ov_Et'], mods['t_resnet'], mods['decoder'], nt_cond, skipco)
    sep_net.eval()
    return sep_net
