"""Layer-shape table of the causal DCCRN (reference: model/causal_netconfig.py:5-103): identical to
net_config except that the encoder pads one frame in time (the causal conv then drops the last frame)."""
from .net_config import _table


def get_net_params():
    return _table(time_pad=1)
