#!/usr/bin/env python3
"""CLI with the reference's surface (reference: tts_service.py): class ``TTS`` with
``load_model / list_voices / load_voice / generate_with_context / generate_audio_segment /
export_wav / say`` and the flags ``-d/--device  -v/--voice  text  --output  --temp  --topk`` (no text: interactive mode).

Only the hot path is re-implemented: generation runs on the MI355X kernels through
``sesameai.generator``.  Host-side audio post-processing that the reference does with pydub
(peak-normalise -> int16, 500 ms lead / 100 ms tail silence, 50 ms fades, tts_service.py:288-306)
is done with NumPy and the stdlib ``wave`` module; playback, watermarking and the WAV/resample
are out of scope.  Voice prompts come from a ``samples.py``-style registry of
WAV files (loaded with ``wave`` + polyphase resampling, encoded by the GPU Mimi encoder) or from
pre-tokenised ``<voice>.pt`` files holding ``[(text | token ids, codes[32,T]), ...]``.
"""
import argparse
import os
import re
import sys
import time
import wave
from typing import List, Optional

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

from sesameai.watermarking import CSM_1B_GH_WATERMARK, load_watermarker, watermark, _resample as resample_to
from sesameai.generator import Segment, load_csm_1b  # noqa: E402  (same import line as the reference, tts_service.py:22)


def discover_voices(voice_dir: str) -> dict:
    """voice name -> prompt source: ``<voice>.pt`` (pre-tokenised ``[(text, codes[32,T]), ...]``) or a
    ``samples.py``-style module next to it defining ``name = {wav_path: transcript}`` dicts
    (reference: samples.py:14-25, tts_service.py:36-42)."""
    voices = {}
    if os.path.isdir(voice_dir):
        for f in sorted(os.listdir(voice_dir)):
            if f.endswith(".pt"):
                voices[os.path.splitext(f)[0]] = os.path.join(voice_dir, f)
        sp = os.path.join(voice_dir, "samples.py")
        if os.path.exists(sp):
            import importlib.util
            spec = importlib.util.spec_from_file_location("csm_voice_samples", sp)     # the reference does `import samples`
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            for name, obj in vars(mod).items():
                if not name.startswith("__") and isinstance(obj, dict):
                    voices[name] = obj
    return voices


def load_audio(path: str, target_rate: int) -> torch.Tensor:
    """reference: TTS._load_audio (tts_service.py:141-168) without torchaudio: PCM WAV via the stdlib
    ``wave`` module, stereo -> mono by averaging, polyphase resampling to the codec rate."""
    with wave.open(path, "rb") as f:
        nch, width, rate, n = f.getnchannels(), f.getsampwidth(), f.getframerate(), f.getnframes()
        raw = f.readframes(n)
    if width == 2:
        x = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
    elif width == 4:
        x = np.frombuffer(raw, dtype="<i4").astype(np.float32) / 2147483648.0
    elif width == 1:
        x = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    else:
        raise ValueError(f"{path}: unsupported sample width {width}")
    if nch > 1:
        x = x.reshape(-1, nch).mean(axis=1)
    if rate != target_rate:
        from math import gcd
        from scipy.signal import resample_poly
        g = gcd(rate, target_rate)
        x = resample_poly(x, target_rate // g, rate // g).astype(np.float32)
    return torch.from_numpy(np.ascontiguousarray(x))


class TTS:
    """reference: tts_service.py:44-525."""

    voice_name = None
    voice_data = None

    def __init__(self, device: str = "cuda", model_repo: str = "sesame/csm-1b", voice_dir: Optional[str] = None) -> None:
        self.device = device
        self.model_repo = model_repo
        self.generator = None
        self.cached_context_tokens: List[torch.Tensor] = []
        self.cached_context_masks: List[torch.Tensor] = []
        self.voices = discover_voices(voice_dir or os.environ.get("CSM_VOICE_DIR", os.path.join(HERE, "voices")))

    def load_model(self) -> None:
        print("Open Sesame...")
        self.generator = load_csm_1b(self.device)
        # reference tts_service.py load_model: the watermarker is loaded next to the model (a pass-through hook
        # that says so once when silentcipher is not installed)
        self.watermarker = load_watermarker(self.device)

    def list_voices(self) -> list:
        return list(self.voices.keys())

    def load_voice(self, voice_name: str) -> None:
        """reference: tts_service.py:105-119."""
        if voice_name not in self.voices:
            raise ValueError(f"Voice '{voice_name}' not found. Available voices: {list(self.voices.keys())}")
        self.cached_context_tokens, self.cached_context_masks = [], []
        self.voice_name = voice_name
        src = self.voices[voice_name]
        self.voice_data = src if isinstance(src, dict) else torch.load(src)      # {wav path: transcript} | [(text, codes[32,T]), ...]
        self._prepare_context()
        self.generate_audio_segment("I'm getting all warmed up for our chatting to begin.")   # warm-up, :119

    def _load_audio(self, audio_path: str) -> torch.Tensor:
        """reference: tts_service.py:141-168 (mono, resampled to the codec's rate)."""
        return load_audio(audio_path, self.generator.sample_rate)

    def _prepare_context(self) -> None:
        """reference: tts_service.py:122-139 -- the voice prompt's segments tokenised once (WAV prompts are encoded by the GPU Mimi encoder;
        pre-tokenised prompts carry their codes) and cached for every sentence; with them cached, Model.prefill_prompt re-runs only the rows
        after the shared prefix."""
        if not self.generator:
            raise ValueError("Model not loaded. Call load_model() first.")
        print(f"Preparing reference audio context for voice: {self.voice_name}...")
        if isinstance(self.voice_data, dict):
            segments = [Segment(speaker=1, text=text, audio=self._load_audio(path)) for path, text in self.voice_data.items()]
        else:
            segments = [Segment(speaker=1, text=text, audio_codes=codes) for text, codes in self.voice_data]
        for segment in segments:
            tokens, masks = self.generator._tokenize_segment(segment)
            self.cached_context_tokens.append(tokens)
            self.cached_context_masks.append(masks)
        print("Reference audio context prepared")

    @torch.inference_mode()
    def generate_with_context(self, prompt, speaker: int = 1, max_audio_length_ms: float = 60_000,
                              temperature: float = 0.9, topk: int = 50) -> torch.Tensor:
        """reference: tts_service.py:170-258, watermark included (:249-256)."""
        g = self.generator
        gen_tokens, gen_masks = g._tokenize_text_segment(prompt, speaker)
        prompt_tokens = torch.cat(self.cached_context_tokens + [gen_tokens], dim=0).long().to(g.device)
        prompt_mask = torch.cat(self.cached_context_masks + [gen_masks], dim=0).bool().to(g.device)
        max_audio_frames = int(max_audio_length_ms / 80)
        max_seq_len = 2048 - max_audio_frames
        if prompt_tokens.size(0) >= max_seq_len:
            raise ValueError(f"Input too long ({prompt_tokens.size(0)} tokens). Maximum is {max_seq_len} tokens.")
        frames = g.generate_codes(prompt_tokens, prompt_mask, max_audio_frames, temperature, topk)
        audio = g._decode_frames(frames)
        if audio.numel() == 0:
            return audio
        # every generated clip carries the public CSM watermark, then goes back to the generator's rate
        audio, wm_rate = watermark(getattr(self, "watermarker", None), audio, g.sample_rate, CSM_1B_GH_WATERMARK)
        return resample_to(audio, wm_rate, g.sample_rate)

    def generate_audio_segment(self, prompt, fade_duration: int = 50, start_silence_duration: int = 500,
                               end_silence_duration: int = 100, temperature: float = 0.8, topk: int = 40) -> np.ndarray:
        """-> int16 mono PCM @ generator.sample_rate (the reference returns a pydub AudioSegment
        holding the same samples, tts_service.py:260-308)."""
        audio = self.generate_with_context(prompt, speaker=1, max_audio_length_ms=30_000, temperature=temperature, topk=topk)
        audio = audio.to(torch.float32).reshape(-1)
        audio = audio / max(float(audio.abs().max()) if audio.numel() else 0.0, 1e-6)
        pcm = (audio.cpu().numpy() * 32767).astype("int16")
        sr = self.generator.sample_rate
        pcm = np.concatenate([np.zeros(sr * start_silence_duration // 1000, np.int16), pcm,
                              np.zeros(sr * end_silence_duration // 1000, np.int16)])
        n = min(sr * fade_duration // 1000, len(pcm) // 2)
        if n > 0:
            ramp = np.linspace(0.0, 1.0, n, dtype=np.float32)
            pcm[:n] = (pcm[:n] * ramp).astype(np.int16)
            pcm[-n:] = (pcm[-n:] * ramp[::-1]).astype(np.int16)
        return pcm

    def _generate_audio_segment_wrapper(self, sentence, fade_duration, start_silence_duration, end_silence_duration, temperature=0.8, topk=40):
        """reference: tts_service.py:310-311."""
        return self.generate_audio_segment(sentence, fade_duration, start_silence_duration, end_silence_duration, temperature, topk)

    def say(self, text: str, output_filename: Optional[str] = "combined_output.wav", fallback_duration: int = 1000,
            fade_duration: int = 50, start_silence_duration: int = 500, end_silence_duration: int = 100,
            temperature: float = 0.8, topk: int = 40, on_segment=None) -> None:
        """reference: tts_service.py:313-470 -- the generation half: the text is split into sentences, each sentence is
        generated against the cached voice context (the shared prompt prefix is prefilled once: Model.prefill_prompt) and
        reported with the reference's line ``> sentence ... [Audio: 1.84s in 0.09s, RTF: 20.44x]``; a sentence that fails is
        replaced by ``fallback_duration`` ms of faded silence; the segments are written to ``output_filename`` if given.
        Playback (the reference's pydub/ffplay player thread) is out of scope: ``on_segment(pcm_int16, sample_rate)``, if
        given, receives every segment the moment it exists -- that is where a player thread's queue would be fed."""
        import textwrap
        sentences = [s for s in re.split(r"(?<=[.!?])\s+", textwrap.dedent(text).strip()) if s.strip()]
        if not sentences:
            print("No valid text to process")
            return
        sr = self.generator.sample_rate
        segments: List[np.ndarray] = []
        for sentence in sentences:
            print(f"> {sentence} ... ", end="", flush=True)
            t0 = time.time()
            try:
                seg = self.generate_audio_segment(sentence, fade_duration=fade_duration, start_silence_duration=start_silence_duration,
                                                  end_silence_duration=end_silence_duration, temperature=temperature, topk=topk)
                took, secs = time.time() - t0, len(seg) / sr
                print(f"[Audio: {secs:.2f}s in {took:.2f}s, RTF: {secs / max(took, 1e-9):.2f}x]")
            except KeyboardInterrupt:
                print("\nExiting due to KeyboardInterrupt")
                break
            except Exception as e:  # noqa: BLE001 -- the reference keeps going with silence
                print(f"Error generating audio for sentence: {sentence}: {e}")
                seg = np.zeros(sr * fallback_duration // 1000, np.int16)
            segments.append(seg)
            if on_segment is not None:
                on_segment(seg, sr)
        if not output_filename:
            return
        if not segments:
            print("No audio segments generated to export")
            return
        combined = np.concatenate(segments)
        with wave.open(output_filename, "wb") as f:
            f.setnchannels(1); f.setsampwidth(2); f.setframerate(sr)
            f.writeframes(combined.tobytes())
        print(f"Export complete: {len(combined) / sr:.2f} seconds of audio")

    def export_wav(self, text: str, output_filename: str, fallback_duration: int = 1000, max_retries: int = 2,
                   temperature: float = 0.8, topk: int = 40) -> None:
        """reference: tts_service.py:472-525."""
        sentences = [s for s in re.split(r"(?<=[.!?])\s+", text) if s.strip()]
        sr = self.generator.sample_rate
        segments = []
        t0 = time.time()
        for sentence in sentences:
            seg, retries = None, 0
            while retries <= max_retries:
                try:
                    print(f"Export: Generating audio for sentence: {sentence} (Attempt {retries + 1})")
                    seg = self.generate_audio_segment(sentence, temperature=temperature, topk=topk)
                    break
                except Exception as e:  # noqa: BLE001 -- same retry policy as the reference
                    retries += 1
                    print(f"Export: Error for sentence: {sentence} (Attempt {retries}): {e}")
            if seg is None:
                print(f"Export: Using fallback for sentence: {sentence}")
                seg = np.zeros(sr * fallback_duration // 1000, np.int16)
            segments.append(seg)
        if not segments:
            print("No audio segments to export")
            return
        combined = np.concatenate(segments)
        with wave.open(output_filename, "wb") as f:
            f.setnchannels(1); f.setsampwidth(2); f.setframerate(sr)
            f.writeframes(combined.tobytes())
        secs = len(combined) / sr
        print(f"Export complete: {secs:.2f} seconds of audio, RTF {secs / max(time.time() - t0, 1e-9):.2f}x")


def main():
    parser = argparse.ArgumentParser(description="SesameAI CSM-1B Text-to-Speech (MI355X build)")
    parser.add_argument("-d", "--device", type=str, default="cuda", help="Device to run on (cuda)")
    parser.add_argument("-v", "--voice", type=str, default=None, help="Voice to use (a <voice>.pt prompt file in the voice dir)")
    parser.add_argument("text", type=str, nargs="?", help="Text to synthesize")
    parser.add_argument("--output", type=str, default="output.wav", help="Output filename")
    parser.add_argument("--temp", "--temperature", type=float, default=0.8, dest="temp")
    parser.add_argument("--topk", type=int, default=40)
    args = parser.parse_args()
    if args.device == "cpu":
        parser.error("this build runs the hot path on MI355X only; there is no -d cpu path (use the reference)")
    tts = TTS(device=args.device)
    tts.load_model()
    if args.voice:
        tts.load_voice(args.voice)
    if args.text:
        tts.export_wav(args.text, args.output, temperature=args.temp, topk=args.topk)
        return
    # reference: tts_service.py:560-572 -- no text: interactive mode.  The reference plays every line; playback is out of
    # scope here, so each line is written to <output stem>_<n>.wav instead.
    print(f"Interactive mode (temp={args.temp}, topk={args.topk})")
    stem, ext = os.path.splitext(args.output)
    n = 0
    while True:
        try:
            line = input("> ")
        except (EOFError, KeyboardInterrupt):
            break
        if line.lower() in ("exit", "quit"):
            break
        if line.strip():
            n += 1
            tts.say(line, output_filename=f"{stem}_{n}{ext or '.wav'}", temperature=args.temp, topk=args.topk)
    print("\nExiting interactive mode.")


if __name__ == "__main__":
    main()
