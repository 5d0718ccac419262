"""MI355X-native I-DCCRN-VAE enhancement hot path (package directory ``i-dccrn-vae_amd``).

Import with ``importlib.import_module("i-dccrn-vae_amd")`` or through the alias module
``idccrn_vae_amd`` at the repository root.
"""
from . import _lib  # noqa: F401
from . import ops  # noqa: F401


def smoke():
    raise NotImplementedError
