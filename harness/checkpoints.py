"""Files the reference's evaluation scripts read back, written in the reference's own formats
and readable by the reference's own ModelManager.

* differentiable quantization (ref: cifar10_test.py:265-270):
      <base>quant_points_<b>bits                     pickle of (points, infoDict) where `points` is a list
                                                     (one entry per quantized tensor) of plain Python
                                                     float lists
      <base>quant_points_<b>bits_model_state_dict    torch.save(quantized model state_dict)
* a ModelManager store (ref: model_manager.py):
      <save_file>                pickle of (name, verbose, {model_name: [(path_model, path_metadata), ...]})
                                 (:321-347; entry 0 is the "creation" entry of add_new_model, :197-219)
      <path><run>                torch.save(model.state_dict())                                     (:182-184)
      <path><run>_metadata       pickle of a LIST of two sanitised dicts (train arguments, infoDict): every value
                                 that is not a number / str / bool / None (or a list/tuple of those) is replaced
                                 by its repr, callables by 'Name: .. Repr: ..'                        (:214-249)
  RunStore writes exactly these, so `ModelManager(save_file)`, `.load_metadata(name, run)`,
  `.load_model_state_dict(name)`, `.get_num_training_runs(name)` of the reference open them
  (tests/test_checkpoints_reference.py does that with the reference's class itself).

The training-run bookkeeping beyond the file formats (continuing from a run, copying histories,
e-mail notifications) is the reference's control plane and stays out of scope.
"""
import numbers
import os
import pickle

import torch

_GOOD = (numbers.Number, str, bool)


def _is_plain(v):
    if isinstance(v, (tuple, list)):
        return all(_is_plain(x) for x in v)
    return isinstance(v, _GOOD) or v is None


def sanitize_metadata(dicts):
    """The picklable form ModelManager.save_metadata stores (ref: model_manager.py:214-249): a list with one
    dict per input dict; plain values are kept, callables become 'Name: <__name__>. Repr: <repr>' (just the repr
    when they have no __name__), everything else its repr."""
    out = []
    for d in dicts:
        if not isinstance(d, dict):
            raise ValueError('Wrong type: the metadata to save should be a tuple of dictionaries')
        clean = {}
        for key, val in d.items():
            if not _is_plain(val):
                if callable(val):
                    try:
                        val = 'Name: {}. Repr: {}'.format(val.__name__, repr(val))
                    except Exception:                       # noqa: BLE001 -- as the reference: any failure -> repr
                        val = repr(val)
                else:
                    val = repr(val)
            clean[key] = val
        out.append(clean)
    return out


def save_metadata(dicts, path):
    with open(path, 'wb') as f:
        pickle.dump(sanitize_metadata(dicts), f)


def _cpu_state_dict(sd):
    return type(sd)((k, v.detach().cpu() if isinstance(v, torch.Tensor) else v) for k, v in sd.items())


class RunStore(object):
    """Writer (and reader) of a ModelManager store: the manager file plus the per-run pairs."""

    def __init__(self, save_file, name=None, verbose=True, create=False):
        self.save_file = save_file
        if create:
            if os.path.exists(save_file):                                            # ref: :30-32
                raise ValueError('The file specified "{}" already exists. Choose another one'.format(save_file))
            if not isinstance(name, str):                                            # ref: :34-35
                raise ValueError('"name" parameter must be a string')
            self.name, self.verbose, self.saved_models = name, verbose, {}
            self.save()
        else:
            with open(save_file, 'rb') as f:                                         # ref: :349-360
                self.name, self.verbose, saved = pickle.load(f)
            self.saved_models = {k: [tuple(x) for x in v] for k, v in saved.items()}

    def save(self):
        """Atomic rewrite of the manager file (ref: :321-347)."""
        obj = (self.name, self.verbose, {k: [tuple(x) for x in v] for k, v in self.saved_models.items()})
        tmp = self.save_file + 'temp'
        with open(tmp, 'wb') as f:
            pickle.dump(obj, f)
        os.replace(tmp, self.save_file)

    def add_new_model(self, model_name, path_to_save, arguments_creator_function=None):
        """Entry 0 of a model's history: an empty weights file and the creator arguments (ref: :197-219)."""
        if not isinstance(model_name, str):
            raise ValueError('model_name parameter must be a string')
        if model_name in self.saved_models:
            raise ValueError('The model name "{}" is already present. Choose a new name'.format(model_name))
        if os.path.exists(path_to_save):
            raise ValueError('The path specified "{}" already exists. Choose a new one'.format(path_to_save))
        with open(path_to_save, 'wb'):
            pass
        meta = path_to_save + '_metadata'
        save_metadata((arguments_creator_function or {}, {}), meta)
        self.saved_models[model_name] = [(path_to_save, meta)]
        self.save()

    def append_run(self, model_name, state_dict, train_arguments, info_dict):
        """What ModelManager.train_model does after training (ref: :160-190): weights at <base><run>, the two
        dicts at <base><run>_metadata, one more history entry, manager file rewritten."""
        if model_name not in self.saved_models:
            raise ValueError('the model_name specified ({}) does not exist in the list of saved models'.format(model_name))
        if info_dict.get('numEpochsTrained', 0) == 0:                                # ref: :162-164
            return None
        run = len(self.saved_models[model_name])
        path = self.saved_models[model_name][0][0] + str(run)
        torch.save(_cpu_state_dict(state_dict), path)
        save_metadata((train_arguments, info_dict), path + '_metadata')
        self.saved_models[model_name].append((path, path + '_metadata'))
        self.save()
        return path

    # readers (same semantics as the reference's, :251-268 / :362-380)
    def get_num_training_runs(self, model_name):
        return len(self.saved_models[model_name]) - 1

    def get_model_base_path(self, model_name):
        return self.saved_models[model_name][0][0]

    def load_metadata(self, model_name, idx_run=-1):
        with open(self.saved_models[model_name][idx_run][1], 'rb') as f:
            return pickle.load(f)

    def load_model_state_dict(self, model_name, idx_run=-1, map_location='cpu'):
        if len(self.saved_models[model_name]) - 1 < 1:
            raise ValueError("The model specified hasn't been trained yet")
        return torch.load(self.saved_models[model_name][idx_run][0], map_location=map_location)


def save_quantization_points(path, points, info_dict, quantized_state_dict):
    """`points`: [ntensors, k] tensor or a list of 1-D tensors (rows padded with +inf -- tensors given fewer
    points by the automatic bit allocation -- are trimmed).  ref: cifar10_test.py:265-270."""
    rows = points if isinstance(points, (list, tuple)) else list(points)
    as_lists = []
    for p in rows:
        p = p.detach().view(1, -1).cpu()
        p = p[:, torch.isfinite(p[0])]
        as_lists.append(p.numpy().tolist()[0])                                           # ref: :265
    with open(path, 'wb') as f:
        pickle.dump((as_lists, dict(info_dict)), f)
    torch.save(_cpu_state_dict(quantized_state_dict), path + '_model_state_dict')
    return path


def load_quantization_points(path, map_location='cpu'):
    with open(path, 'rb') as f:
        points, info = pickle.load(f)
    return points, info, torch.load(path + '_model_state_dict', map_location=map_location)
