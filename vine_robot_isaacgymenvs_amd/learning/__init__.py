"""PPO rollout/update for the Vine5LinkMovingBase path (rows R1-R6 of SURVEY 8a).

rl-games==1.5.2 (reference setup.py:22) is not vendored and not installed; this package restates the
part of it that ``cfg/train/Vine5LinkMovingBasePPO.yaml`` selects (``a2c_continuous`` + ``continuous_a2c_logstd``
+ ``actor_critic`` network with LSTM), following the in-tree text of the same arithmetic at
``isaacgymenvs/learning/common_agent.py`` and rl_games' published behaviour (parity unpinned).
"""
