"""Watermark hook with the reference's names (reference: sesameai/watermarking.py:9,20-55), so that
``from sesameai.watermarking import CSM_1B_GH_WATERMARK, watermark`` (reference tts_service.py:23) keeps
working when this package shadows the reference's.

The watermark itself is a third-party model (``silentcipher``, weights fetched from the hub) and is out of
scope of the hot path.  When ``silentcipher`` is importable it is used exactly where the
reference uses it, with SciPy polyphase resampling instead of torchaudio; when it is not, ``load_watermarker``
returns ``None`` and ``watermark`` hands the audio back unchanged (and says so once).
"""
from math import gcd
from typing import List, Optional, Tuple

import torch

# Same public (hence not secret) key as the reference; use a private one in another application.
CSM_1B_GH_WATERMARK = [212, 211, 146, 56, 201]

_WM_RATE = 44100
_warned = False


def _resample(x: torch.Tensor, src: int, dst: int) -> torch.Tensor:
    if src == dst:
        return x
    from scipy.signal import resample_poly
    g = gcd(src, dst)
    y = resample_poly(x.detach().to(torch.float32).cpu().numpy(), dst // g, src // g, axis=-1)
    return torch.from_numpy(y.astype("float32")).to(x.device)


def load_watermarker(device: str = "cuda"):
    try:
        import silentcipher
    except ImportError:
        return None
    return silentcipher.get_model(model_type="44.1k", device=device)


@torch.inference_mode()
def watermark(watermarker, audio_array: torch.Tensor, sample_rate: int, watermark_key: List[int]) -> Tuple[torch.Tensor, int]:
    """-> (audio, sample rate of the returned audio)."""
    global _warned
    if watermarker is None:
        if not _warned:
            print("sesameai.watermarking: silentcipher is not installed; audio is returned without a watermark")
            _warned = True
        return audio_array, sample_rate
    wide = _resample(audio_array, sample_rate, _WM_RATE)
    marked, _ = watermarker.encode_wav(wide, _WM_RATE, watermark_key, calc_sdr=False, message_sdr=36)
    out_rate = min(_WM_RATE, sample_rate)
    return _resample(marked, _WM_RATE, out_rate), out_rate


@torch.inference_mode()
def verify(watermarker, watermarked_audio: torch.Tensor, sample_rate: int, watermark_key: List[int]) -> bool:
    if watermarker is None:
        return False
    res = watermarker.decode_wav(_resample(watermarked_audio, sample_rate, _WM_RATE), _WM_RATE, phase_shift_decoding=True)
    return bool(res["status"]) and res["messages"][0] == watermark_key
