"""Layer-shape table of the non-causal DCCRN (reference: model/net_config.py:5-103).
Same keys and values; built from the channel base instead of spelled out."""


def _table(time_pad: int, base: int = 32) -> dict:
    widths = [1, 2, 4, 4, 8, 8]
    enc = [1] + [base * w for w in widths]
    dec = [base * w for w in reversed(widths)] + [1]
    dec[0] = base * 8
    n = len(widths)
    freq = [129, 65, 33, 17, 9, 5]
    frames = [1600 - i for i in range(n)]          # stale values kept from the reference; only feed unused BN args
    params = {
        "encoder_channels": enc,
        "encoder_kernel_sizes": [(5, 2)] * n,
        "encoder_strides": [(2, 1)] * n,
        "encoder_paddings": [(2, time_pad)] * n,
        "lstm_dim": [base * 8 * 5, 128],
        "dense": [128, base * 8 * 5],
        "lstm_layer_num": 2,
        "decoder_channels": dec,
        "decoder_kernel_sizes": [(5, 2)] * n,
        "decoder_strides": [(2, 1)] * n,
        "decoder_paddings": [(2, 0)] * n,
        "encoder_chw": [(enc[i + 1], freq[i], frames[i]) for i in range(n)],
    }
    dfreq = [9, 17, 33, 65, 129, 257]
    dframes = [1596 + i for i in range(n)]
    params["decoder_chw"] = [(dec[i + 1], dfreq[i], dframes[i]) for i in range(n)]
    return params


def get_net_params():
    return _table(time_pad=0)
