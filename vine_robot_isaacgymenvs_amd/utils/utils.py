"""Seeding helper with the reference's semantics (isaacgymenvs/utils/utils.py:43-71)."""
import os
import random

import numpy as np
import torch


def set_np_formatting():
    np.set_printoptions(edgeitems=30, infstr="inf", linewidth=4000, nanstr="nan", precision=2, suppress=False,
                        threshold=10000, formatter=None)


def set_seed(seed, torch_deterministic=False, rank=0):
    """seed == -1 -> 42 + rank when deterministic, else random; otherwise seed + rank.
    NB train.py adds the rank once more before calling this (train.py:78), so rank r trains with 42 + 2r."""
    if seed == -1 and torch_deterministic:
        seed = 42 + rank
    elif seed == -1:
        seed = np.random.randint(0, 10000)
    else:
        seed = seed + rank
    print("Setting seed: {}".format(seed))
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    if torch_deterministic:
        torch.backends.cudnn.benchmark = False
        torch.backends.cudnn.deterministic = True
        torch.use_deterministic_algorithms(True, warn_only=True)
    else:
        torch.backends.cudnn.benchmark = True
        torch.backends.cudnn.deterministic = False
    return seed
