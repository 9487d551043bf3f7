from .ba import BA, neighbors, reproject
