"""Importable alias of the package directory ``i-dccrn-vae_amd`` (a hyphen is not a valid
identifier, so ``import idccrn_vae_amd`` resolves to it)."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
sys.modules[__name__] = importlib.import_module("i-dccrn-vae_amd")
