"""Files the reference's evaluation scripts read back, written in the reference's own formats.

* differentiable quantization (ref: cifar10_test.py:265-270):
      <base>quant_points_<b>bits                     pickle of (points, infoDict) where `points` is a list
                                                     (one entry per quantized tensor) of plain Python
                                                     float lists
      <base>quant_points_<b>bits_model_state_dict    torch.save(quantized model state_dict)
* a training run of ModelManager (ref: model_manager.py:182-190): torch.save(model.state_dict()) at
  <path><run> and a pickle of (train arguments, infoDict) at <path><run>_metadata.

Only the formats are reproduced; ModelManager itself (run history bookkeeping) is out of scope.
"""
import pickle

import torch


def save_quantization_points(path, points, info_dict, quantized_state_dict):
    """`points`: [ntensors, k] tensor or a list of 1-D tensors."""
    rows = points if isinstance(points, (list, tuple)) else list(points)
    as_lists = [p.detach().view(1, -1).cpu().numpy().tolist()[0] for p in rows]     # ref: :265
    with open(path, 'wb') as f:
        pickle.dump((as_lists, dict(info_dict)), f)
    torch.save({k: v.detach().cpu() for k, v in quantized_state_dict.items()}, path + '_model_state_dict')
    return path


def load_quantization_points(path, map_location='cpu'):
    with open(path, 'rb') as f:
        points, info = pickle.load(f)
    return points, info, torch.load(path + '_model_state_dict', map_location=map_location)


def save_training_run(path, model, train_arguments, info_dict):
    """ModelManager's per-run pair of files (ref: model_manager.py:182-190)."""
    torch.save(model.state_dict(), path)
    with open(path + '_metadata', 'wb') as f:
        pickle.dump((dict(train_arguments), dict(info_dict)), f)
    return path


def load_training_run(path, map_location='cpu'):
    with open(path + '_metadata', 'rb') as f:
        args, info = pickle.load(f)
    return torch.load(path, map_location=map_location), args, info
