"""`import dpvo...` compatibility: the reference's scripts (demo.py, evaluate_*.py) do `from dpvo.dpvo import DPVO`,
`from dpvo.config import cfg`, `from dpvo.utils import Timer`, `from dpvo.plot_utils import ...`.  `install()` registers this
package's modules under the reference's names, so those import lines resolve to the MI355X implementation:

    import dpvo_amd.compat; dpvo_amd.compat.install()
    from dpvo.dpvo import DPVO          # -> dpvo_amd.dpvo.DPVO
"""
import importlib
import sys

_MODULES = ("dpvo", "net", "patchgraph", "config", "utils", "projective_ops", "altcorr", "fastba", "lietorch", "extractor",
            "stream", "plot_utils")


def install(name="dpvo", force=False):
    if name in sys.modules and not force and getattr(sys.modules[name], "__dpvo_amd__", False) is False:
        raise ImportError(f"a different package named {name!r} is already imported")
    pkg = importlib.import_module("dpvo_amd")
    pkg.__dpvo_amd__ = True
    sys.modules[name] = pkg
    for m in _MODULES:
        sys.modules[f"{name}.{m}"] = importlib.import_module(f"dpvo_amd.{m}")
    return pkg
