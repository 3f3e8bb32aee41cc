#!/usr/bin/env python3
"""Drop-in for ``python train.py task=Vine5LinkMovingBase ...`` of the reference (isaacgymenvs/train.py)."""
from vine_robot_isaacgymenvs_amd.train import main

if __name__ == "__main__":
    main()
