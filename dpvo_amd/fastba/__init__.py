from .ba import BA, neighbors, reproject, solve_system
