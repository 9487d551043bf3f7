from .correlation import corr, corr_pyramid, patchify
