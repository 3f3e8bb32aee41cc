"""Process-wide seeding with the reference's rule (isaacgymenvs/utils/utils.py:43-71 ``set_seed``):

    seed == -1 and deterministic  -> 42 + rank
    seed == -1                    -> random in [0, 10000)
    otherwise                     -> seed + rank

train.py adds the rank to cfg.seed *before* calling this (train.py:78), so rank r ends up at 42 + 2r with the
default seed; bench.py and the multi-GPU tests rely on that number.
"""
import os
import random

import numpy as np
import torch


def set_np_formatting():
    np.set_printoptions(precision=2, linewidth=4000, threshold=10000, edgeitems=30, suppress=False)


def resolve_seed(seed, torch_deterministic=False, rank=0):
    if seed != -1:
        return seed + rank
    return 42 + rank if torch_deterministic else int(np.random.randint(0, 10000))


def set_seed(seed, torch_deterministic=False, rank=0):
    seed = resolve_seed(seed, torch_deterministic, rank)
    print(f"Setting seed: {seed}")
    for seeder in (random.seed, np.random.seed, torch.manual_seed):
        seeder(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    # MIOpen/hipBLASLt autotuning on unless bit-reproducibility was requested
    torch.backends.cudnn.benchmark = not torch_deterministic
    torch.backends.cudnn.deterministic = bool(torch_deterministic)
    if torch_deterministic:
        torch.use_deterministic_algorithms(True, warn_only=True)
    return seed
