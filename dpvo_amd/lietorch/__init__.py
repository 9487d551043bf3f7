from .groups import SE3, LieGroup, cat, stack
