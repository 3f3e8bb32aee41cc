"""Task registry, same shape as the reference's ``isaacgymenvs/tasks/__init__.py:37-62`` restricted to
the one task this build accelerates."""
from .vine5link_moving_base import Vine5LinkMovingBase

isaacgym_task_map = {
    "Vine5LinkMovingBase": Vine5LinkMovingBase,
}
