"""Generation driver with the reference's surface (reference: sesameai/generator.py).

``Segment``, ``Generator.generate`` / ``generate_stream`` / ``_tokenize_*`` and ``load_csm_1b``
keep the reference's names, argument meaning and error behaviour; the frame loop itself runs on
the GPU from on-device state (one hipGraph replay per 80 ms frame) and the host only polls the
EOS flag every few frames instead of synchronising on every frame (generator.py:285).
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Callable, Generator as PyGenerator, List, Optional, Sequence, Tuple, Union

import torch

from .models import Model, ModelArgs, csm_1b_args

FRAME_MS = 80                     # generator.py:257
MAX_SEQ_LEN = 2048                # generator.py:276


@dataclass
class Segment:
    """reference: sesameai/generator.py:16-21.  ``audio`` is (num_samples,) @ 24 kHz.
    Extension: ``audio_codes`` (32, T) may carry pre-computed Mimi codes, and ``text`` may be
    a list of token ids, so prompts can be built without the gated tokenizer / Mimi encoder."""
    speaker: int
    text: Union[str, Sequence[int]]
    audio: Optional[torch.Tensor] = None
    audio_codes: Optional[torch.Tensor] = None


def load_llama3_tokenizer(path: Optional[str] = None):
    """reference: sesameai/generator.py:24-38 -- Llama-3.2-1B tokenizer with the bos/eos
    TemplateProcessing.  Loads a local ``tokenizer.json`` (no hub access here)."""
    from tokenizers import Tokenizer
    from tokenizers.processors import TemplateProcessing
    path = path or os.environ.get("CSM_TOKENIZER_JSON")
    if not path or not os.path.exists(path):
        return None
    tok = Tokenizer.from_file(path)
    bos, eos = "<|begin_of_text|>", "<|end_of_text|>"
    bos_id, eos_id = tok.token_to_id(bos), tok.token_to_id(eos)
    tok.post_processor = TemplateProcessing(
        single=f"{bos}:0 $A:0 {eos}:0",
        pair=f"{bos}:0 $A:0 {eos}:0 {bos}:1 $B:1 {eos}:1",
        special_tokens=[(bos, bos_id), (eos, eos_id)])
    return tok


class Generator:
    """reference: sesameai/generator.py:41-300."""

    def __init__(self, model: Model, audio_tokenizer=None, text_tokenizer=None, max_batch_size: int = 1):
        self._model = model
        self._model.setup_caches(max_batch_size)
        self._text_tokenizer = text_tokenizer if text_tokenizer is not None else load_llama3_tokenizer()
        self._audio_tokenizer = audio_tokenizer
        self.sample_rate = getattr(audio_tokenizer, "sample_rate", 24_000)
        self.device = model.device
        self._stream_buffer_size = 10          # generator.py:61
        self._eos_poll = 8                     # frames launched between EOS polls
        self._mimi_stream = None               # side HIP stream for Mimi in generate_stream

    # -- prompt assembly (generator.py:63-109) ------------------------------------------------
    def _text_ids(self, text: Union[str, Sequence[int]], speaker: int) -> List[int]:
        if not isinstance(text, str):
            return [int(t) for t in text]
        if self._text_tokenizer is None:
            raise RuntimeError("no text tokenizer: set CSM_TOKENIZER_JSON to a local Llama-3.2 tokenizer.json "
                               "or pass token ids instead of a string")
        enc = self._text_tokenizer.encode(f"[{speaker}]{text}")
        return list(enc.ids if hasattr(enc, "ids") else enc)

    def _tokenize_text_segment(self, text, speaker: int) -> Tuple[torch.Tensor, torch.Tensor]:
        ids = self._text_ids(text, speaker)
        frame = torch.zeros(len(ids), 33).long()
        mask = torch.zeros(len(ids), 33).bool()
        frame[:, -1] = torch.tensor(ids, dtype=torch.long)
        mask[:, -1] = True
        return frame.to(self.device), mask.to(self.device)

    def _tokenize_audio(self, audio: Optional[torch.Tensor], codes: Optional[torch.Tensor] = None):
        if codes is None:
            assert audio is not None and audio.ndim == 1, "Audio must be single channel"
            if self._audio_tokenizer is None or not hasattr(self._audio_tokenizer, "encode"):
                raise RuntimeError("no Mimi encoder available: pass Segment.audio_codes")
            codes = self._audio_tokenizer.encode(audio.to(self.device).unsqueeze(0).unsqueeze(0))[0]
        codes = codes.to(self.device).long()
        eos = torch.zeros(codes.size(0), 1, dtype=torch.long, device=self.device)     # all-zero EOS frame
        codes = torch.cat([codes, eos], dim=1)
        frame = torch.zeros(codes.size(1), 33, dtype=torch.long, device=self.device)
        mask = torch.zeros(codes.size(1), 33, dtype=torch.bool, device=self.device)
        frame[:, :-1] = codes.transpose(0, 1)
        mask[:, :-1] = True
        return frame, mask

    def _tokenize_segment(self, segment: Segment) -> Tuple[torch.Tensor, torch.Tensor]:
        t, tm = self._tokenize_text_segment(segment.text, segment.speaker)
        a, am = self._tokenize_audio(segment.audio, segment.audio_codes)
        return torch.cat([t, a], dim=0), torch.cat([tm, am], dim=0)

    def _build_prompt(self, text, speaker: int, context: List[Segment]):
        toks, masks = [], []
        for seg in context:
            t, m = self._tokenize_segment(seg)
            toks.append(t); masks.append(m)
        t, m = self._tokenize_text_segment(text, speaker)
        toks.append(t); masks.append(m)
        return torch.cat(toks, 0).long().to(self.device), torch.cat(masks, 0).bool().to(self.device)

    # -- the frame loop -----------------------------------------------------------------------
    def _frame_blocks(self, prompt_tokens: torch.Tensor, prompt_mask: torch.Tensor, max_generation_len: int,
                      temperature: float, topk: int, poll: int) -> PyGenerator[torch.Tensor, None, None]:
        """Yields the generated frames in blocks [n][B][32] int32 (CPU) of about ``poll`` frames, cut at EOS for
        B == 1.  The NEXT block's frame steps are already enqueued on the GPU when a block is yielded, so whatever
        the consumer does with it (Mimi decode on another stream, playback) overlaps the language model."""
        B = prompt_tokens.shape[0]
        m = self._model
        m.reset_caches()
        self.last_eos_at = torch.full((B,), -1, dtype=torch.int32)
        if max_generation_len <= 0:
            return
        m.prefill_prompt(prompt_tokens, prompt_mask)        # reuses the cached KV of a shared voice-prompt prefix
        m.depth(B, temperature, topk, commit=True)
        launched, delivered = 1, 0

        def enqueue() -> int:
            n = min(poll, max_generation_len - launched)
            for _ in range(n):
                m.step(B, temperature, topk)
            return n

        launched += enqueue()
        while True:
            upto = launched
            fr, eos = m.read_frames(B, delivered, upto - delivered)          # waits for the frames launched so far
            self.last_eos_at = eos
            done = bool((eos >= 0).all()) or launched >= max_generation_len
            if not done:
                launched += enqueue()                                       # keep the GPU busy before handing out
            if B == 1 and eos[0] >= 0:
                fr = fr[: max(int(eos[0]) - delivered, 0)]
            delivered = upto
            if fr.shape[0]:
                yield fr
            if done:
                return

    @torch.inference_mode()
    def generate_codes(self, prompt_tokens: torch.Tensor, prompt_mask: torch.Tensor, max_generation_len: int,
                       temperature: float, topk: int, on_frames: Optional[Callable[[torch.Tensor], None]] = None,
                       poll: Optional[int] = None) -> torch.Tensor:
        """prompt (S,33) or (B,S,33) -> frames [n][B][32] int32 (CPU), cut at each sequence's EOS
        for B == 1 (the reference is batch-1: generator.py:47).  For B > 1 all sequences run
        ``max_generation_len`` frames unless every one hit EOS; the caller trims with ``last_eos_at``."""
        if prompt_tokens.dim() == 2:
            prompt_tokens, prompt_mask = prompt_tokens.unsqueeze(0), prompt_mask.unsqueeze(0)
        B, S, _ = prompt_tokens.shape
        max_context_len = MAX_SEQ_LEN - max_generation_len
        if S >= max_context_len:
            raise ValueError(f"Inputs too long, must be below max_seq_len - max_generation_len: {max_context_len}")
        blocks = []
        for fr in self._frame_blocks(prompt_tokens, prompt_mask, max_generation_len, temperature, topk, poll or self._eos_poll):
            if on_frames is not None:
                on_frames(fr)
            blocks.append(fr)
        return torch.cat(blocks) if blocks else torch.empty(0, B, 32, dtype=torch.int32)

    def _decode_frames(self, frames: torch.Tensor) -> torch.Tensor:
        """frames [n][1][32] -> audio (n*1920,) (reference: _decode_frames, generator.py:111-117)."""
        if frames.shape[0] == 0:
            return torch.tensor([])
        if self._audio_tokenizer is None:
            raise RuntimeError("no Mimi decoder attached to this Generator")
        codes = frames.to(self.device).permute(1, 2, 0).contiguous()          # (B, 32, T)
        return self._audio_tokenizer.decode(codes).squeeze(0).squeeze(0)

    def generate_stream(self, text, speaker: int, context: List[Segment], max_audio_length_ms: float = 90_000,
                        temperature: float = 0.7, topk: int = 30,
                        on_chunk_generated: Optional[Callable[[torch.Tensor], None]] = None
                        ) -> PyGenerator[torch.Tensor, None, None]:
        """reference: generator.py:119-210 -- yields audio every ``_stream_buffer_size`` frames as soon as it exists,
        each buffer decoded statelessly like the reference.  Mimi runs on its own HIP stream while the frame steps
        of the next buffer (already enqueued) run on the caller's stream."""
        max_generation_len = int(max_audio_length_ms / FRAME_MS)
        with torch.inference_mode():
            tokens, mask = self._build_prompt(text, speaker, context)
        if tokens.dim() == 2:
            tokens, mask = tokens.unsqueeze(0), mask.unsqueeze(0)
        if tokens.shape[1] >= MAX_SEQ_LEN - max_generation_len:
            raise ValueError(f"Inputs too long, must be below max_seq_len - max_generation_len: {MAX_SEQ_LEN - max_generation_len}")
        on_gpu = torch.device(self.device).type == "cuda"           # (a CPU device only occurs in the host-logic tests)
        if on_gpu and getattr(self, "_mimi_stream", None) is None:
            self._mimi_stream = torch.cuda.Stream(device=self.device)
        side = self._mimi_stream if on_gpu else None
        pending: List[torch.Tensor] = []
        size = self._stream_buffer_size

        def decode(n: int) -> torch.Tensor:
            buf = torch.stack(pending[:n]); del pending[:n]
            if side is None:
                with torch.inference_mode():
                    return self._decode_frames(buf)
            with torch.inference_mode(), torch.cuda.stream(side):
                pcm = self._decode_frames(buf)
            side.synchronize()
            return pcm

        blocks = self._frame_blocks(tokens, mask, max_generation_len, temperature, topk, size)
        while True:
            with torch.inference_mode():
                fr = next(blocks, None)
            if fr is None:
                break
            pending.extend(fr.unbind(0))
            while len(pending) >= size:
                chunk = decode(size)
                if on_chunk_generated:
                    on_chunk_generated(chunk)
                yield chunk
        if pending:
            chunk = decode(len(pending))
            if on_chunk_generated:
                on_chunk_generated(chunk)
            yield chunk

    @torch.inference_mode()
    def generate(self, text, speaker: int, context: List[Segment], max_audio_length_ms: float = 90_000,
                 temperature: float = 0.7, topk: int = 30, stream: bool = False) -> torch.Tensor:
        """reference: generator.py:212-300."""
        if stream:
            chunks = list(self.generate_stream(text, speaker, context, max_audio_length_ms, temperature, topk))
            return torch.cat(chunks) if chunks else torch.tensor([])
        max_generation_len = int(max_audio_length_ms / FRAME_MS)
        tokens, mask = self._build_prompt(text, speaker, context)
        frames = self.generate_codes(tokens, mask, max_generation_len, temperature, topk)
        if frames.shape[0] == 0:
            return torch.tensor([])
        return self._decode_frames(frames)


def load_csm_1b(device: str = "cuda", model_path: Optional[str] = None, mimi_path: Optional[str] = None,
                max_batch_size: int = 1, synthetic: Optional[bool] = None) -> Generator:
    """reference: sesameai/generator.py:330-346.  The reference downloads ``sesame/csm-1b`` and the Mimi
    checkpoint from the hub; there is no network here, so ``model_path`` / ``mimi_path`` (or $CSM_MODEL_PATH /
    $CSM_MIMI_PATH) must name local files.  Seeded random weights of the true shapes (benchmarks, tests)
    are used ONLY when asked for -- ``synthetic=True`` or $CSM_SYNTHETIC=1 -- never as a silent fallback:
    a TTS service that writes noise and exits 0 is worse than one that refuses to start."""
    model_path = model_path or os.environ.get("CSM_MODEL_PATH")
    mimi_path = mimi_path or os.environ.get("CSM_MIMI_PATH")
    if synthetic is None:
        synthetic = os.environ.get("CSM_SYNTHETIC") == "1"
    if not synthetic and not (model_path and mimi_path):
        raise FileNotFoundError("load_csm_1b: set CSM_MODEL_PATH (sesame/csm-1b model.safetensors) and CSM_MIMI_PATH (moshi "
                                "tokenizer safetensors) or pass model_path= / mimi_path=; for seeded random weights "
                                "(benchmarks, tests) pass synthetic=True or set CSM_SYNTHETIC=1")
    model = Model.from_pretrained(model_path, device=device) if model_path else Model(csm_1b_args(), None, device=device)
    from .mimi import MimiCodec
    mimi = MimiCodec.from_pretrained(mimi_path, device=device)
    return Generator(model, audio_tokenizer=mimi, max_batch_size=max_batch_size)
