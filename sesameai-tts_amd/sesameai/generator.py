"""Generation driver with the reference's surface (reference: sesameai/generator.py).

``Segment``, ``Generator.generate`` / ``generate_stream`` / ``_tokenize_*`` and ``load_csm_1b``
keep the reference's names, argument meaning and error behaviour; the frame loop itself runs on
the GPU from on-device state (one hipGraph replay per 80 ms frame) and the host only polls the
EOS flag every few frames instead of synchronising on every frame (generator.py:285).
"""
from __future__ import annotations

import os
import queue
import struct
import threading
import time
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


class _FirstBlockGate:
    """Handed by ``generate_stream`` to ``_frame_blocks``: the frame loop binds the action that queues the block after the
    first, the stream calls ``release()`` once the first chunk's decode has been submitted.  One gate per stream -- two
    streams on one Generator cannot release each other's blocks, and an abandoned stream leaves nothing installed."""

    def __init__(self) -> None:
        self._action: Optional[Callable[[], None]] = None

    def bind(self, action: Optional[Callable[[], None]]) -> None:
        self._action = action

    def release(self) -> None:
        action, self._action = self._action, None
        if action is not None:
            action()


class Generator:
    """reference: sesameai/generator.py:41-300."""

    def __init__(self, model: Model, audio_tokenizer=None, text_tokenizer=None, max_batch_size: int = 1):
        self._model = model
        self._model.setup_caches(max_batch_size)
        self._max_batch = max_batch_size
        self._text_tokenizer = text_tokenizer if text_tokenizer is not None else load_llama3_tokenizer()
        self._audio_tokenizer = audio_tokenizer
        self.sample_rate = getattr(audio_tokenizer, "sample_rate", 24_000)
        self.device = model.device
        self._stream_buffer_size = 10          # generator.py:61
        self._eos_poll = 8                     # frames launched between EOS polls
        self._mimi_stream = None               # side HIP stream for Mimi in generate_stream

    def warm_up(self, temperature: float = 0.7, topk: int = 30, also=((0.8, 40), (0.9, 50))) -> None:
        """One short synthetic utterance through the streaming path and the whole-utterance path.  Everything a process does once --
        torch's first device kernels of each kind, the HIP module load, the frame-step graph's capture (for THIS temperature / top-k),
        the Mimi chunk graph, the side stream -- is paid here instead of by the first request: measured on an MI355X
        (tools/dbg/first_chunk_breakdown.py), the first utterance of a process took 164 ms to its first chunk against 30 ms for
        every later one.  ``load_csm_1b`` calls it (CSM_NO_WARMUP=1 skips it); the reference has no counterpart.
        ``also``: further (temperature, top-k) pairs whose frame-step graphs are captured too -- the engine keeps 4 captured steps per
        handle (csm_frame_step), and the reference's callers use 0.8 / 40 (tts_service.py:266) and 0.9 / 50 (:175) beside 0.7 / 30."""
        if self._audio_tokenizer is None or self.device.type != "cuda":
            return
        g = torch.Generator().manual_seed(0)
        ctx = [Segment(speaker=0, text=[11] * 40, audio_codes=torch.randint(0, 2048, (32, 60), generator=g))]
        n = 2 * self._stream_buffer_size + 2
        for _ in self.generate_stream([11] * 12, 0, ctx, max_audio_length_ms=n * FRAME_MS, temperature=temperature, topk=topk):
            pass
        self.generate([11] * 12, 0, ctx, max_audio_length_ms=4 * FRAME_MS, temperature=temperature, topk=topk)
        for (t, k) in also or ():
            if (float(t), int(k)) != (float(temperature), int(topk)):
                self._model.step(1, float(t), int(k))       # one more frame on the warm-up's state: captures that key's graph
        self._model._kv_prompt = None                       # the synthetic prompt is nobody's prefix
        # the warm-up's frames advanced the Philox step counter: put the noise stream back where a process without the warm-up
        # (CSM_NO_WARMUP=1) has it, so the two produce the same audio for the same requests
        self._model.seed(getattr(self._model, "_seed_value", 0))
        torch.cuda.synchronize(self.device)

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
                      temperature: float, topk: int, poll: int, gate: Optional["_FirstBlockGate"] = None
                      ) -> PyGenerator[torch.Tensor, None, None]:
        """Yields the generated frames in blocks [n][B][32] int32 (CPU) of about ``poll`` frames, cut at EOS for
        B == 1.  The NEXT block's frame steps are already enqueued on the GPU when a block is yielded, so whatever
        the consumer does with it (Mimi decode on another stream, playback) overlaps the language model.

        ``gate`` (streaming only): time to the first audio.  With a gate the FIRST block (frame 0 + poll - 1 steps) is handed
        out before anything else is queued -- its consumer (the first Mimi decode) then has the GPU to itself instead of
        squeezing between frame steps whose persistent launches occupy every CU (first 10-frame chunk: 38.8 -> 31 ms) -- and
        the second block is queued when the consumer calls ``gate.release()`` (generate_stream: right after submitting the
        first decode, BEFORE the chunk goes to the user) or, at the latest, when it asks for the next block.  Without a gate
        (generate_codes: nobody decodes between blocks) every block, the first included, is followed at once by the next."""
        B = prompt_tokens.shape[0]
        m = self._model
        m.reset_caches()
        self.last_eos_at = torch.full((B,), -1, dtype=torch.int32)
        if max_generation_len <= 0:
            return
        m.prefill_prompt(prompt_tokens, prompt_mask)        # reuses the cached KV of a shared voice-prompt prefix
        m.depth(B, temperature, topk, commit=True)
        launched, delivered = 1, 0

        def enqueue(want: int) -> int:
            n = max(min(want, max_generation_len - launched), 0)
            for _ in range(n):
                m.step(B, temperature, topk)
            return n

        launched += enqueue(poll - 1)
        held = gate is not None                 # the block after the first waits for the consumer's go

        def open_gate() -> None:
            nonlocal launched, held
            if held:
                held = False
                launched += enqueue(poll)

        if gate is not None:
            gate.bind(open_gate)
        try:
            while True:
                upto = launched
                fr, eos = m.read_frames(B, delivered, upto - delivered)          # waits for the frames launched so far
                self.last_eos_at = eos
                done = bool((eos >= 0).all()) or launched >= max_generation_len
                if done:
                    held = False                                                # nothing more to launch
                elif not held:
                    launched += enqueue(poll)                                   # keep the GPU busy before handing out
                if B == 1 and eos[0] >= 0:
                    fr = fr[: max(int(eos[0]) - delivered, 0)]
                delivered = upto
                if fr.shape[0]:
                    yield fr
                if done:
                    return
                open_gate()                                                     # a consumer that never released is released now
        finally:
            held = False
            if gate is not None:
                gate.bind(None)

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

    @torch.inference_mode()
    def iter_codes_continuous(self, prompts: Sequence[Tuple[torch.Tensor, torch.Tensor]], max_generation_len: Union[int, Sequence[int]],
                              temperature: float, topk: int, poll: Optional[int] = None
                              ) -> PyGenerator[Tuple[int, torch.Tensor], None, None]:
        """Any number of prompts [(tokens (S_i,33), mask (S_i,33)), ...] of any lengths through a batch of ``max_batch_size``
        slots that is kept FULL: an utterance that reaches its all-zero EOS frame (generator.py:285) or the length limit is
        retired and its slot re-prefilled with the next prompt (Model.refill_slot) while the other slots keep generating --
        their frames are bit-identical to an undisturbed run.  Yields ``(index of the prompt, frames [n_i][32] int32 CPU)`` as
        each utterance FINISHES (cut at its EOS like the reference's batch-1 loop, or at ``max_generation_len`` -- one limit for
        all prompts or one per prompt), so a long run hands its results out as it
        goes.  There is no limit on the total number of frame steps: the engine's frame history is a ring (include/csm_hip.h,
        csm_read_frames) and every block of ``poll`` steps is read before the next one is launched."""
        from collections import deque
        m = self._model
        # one length limit for all, or one per prompt (a request's own max_audio_length_ms)
        limits = [int(max_generation_len)] * len(prompts) if isinstance(max_generation_len, (int, float)) else [int(x) for x in max_generation_len]
        if len(limits) != len(prompts):
            raise ValueError("max_generation_len: one value, or one per prompt")
        for (t, _), lim in zip(prompts, limits):
            if t.shape[0] >= MAX_SEQ_LEN - lim:
                raise ValueError(f"Inputs too long, must be below max_seq_len - max_generation_len: {MAX_SEQ_LEN - lim}")
        if not prompts:
            return
        poll = poll or self._eos_poll
        B = min(self._max_batch, len(prompts))
        beside = getattr(m, "supports_refill_beside_the_loop", None)
        if beside is not None and beside(B) and getattr(self, "refill_beside_the_loop", True):
            yield from self._iter_codes_refilling_beside_the_loop(prompts, limits, temperature, topk, poll, B)
            return
        pending = deque(range(len(prompts)))
        slot_idx: List[Optional[int]] = [None] * B
        slot_frames: List[List[torch.Tensor]] = [[] for _ in range(B)]
        empty = torch.empty(0, 32, dtype=torch.int32)
        finished: List[Tuple[int, torch.Tensor]] = []
        m.reset_caches()

        def start(slot: int) -> bool:
            while pending:
                i = pending.popleft()
                t, mk = prompts[i]
                f0 = m.refill_slot(slot, t, mk, temperature, topk).cpu()
                if limits[i] <= 0 or bool((f0 == 0).all()):
                    finished.append((i, empty))                         # EOS in the very first frame: empty utterance (generator.py:296)
                    continue
                slot_idx[slot], slot_frames[slot] = i, [f0]
                return True
            slot_idx[slot] = None
            return False

        for s_ in range(B):
            start(s_)
        yield from finished
        finished.clear()
        g = m.num_frames()                                              # next global frame index
        while any(i is not None for i in slot_idx):
            active = [s_ for s_ in range(B) if slot_idx[s_] is not None]
            done = [s_ for s_ in active if len(slot_frames[s_]) >= limits[slot_idx[s_]]]
            if not done:
                n = min(poll, min(limits[slot_idx[s_]] - len(slot_frames[s_]) for s_ in active))
                for _ in range(n):
                    m.step(B, temperature, topk)
                fr, eos = m.read_frames(B, g, n)
                for s_ in active:
                    rows = fr[:, s_]
                    if int(eos[s_]) >= 0:
                        rows = rows[: max(int(eos[s_]) - g, 0)]
                        done.append(s_)
                    slot_frames[s_].extend(rows.unbind(0))
                    if len(slot_frames[s_]) >= limits[slot_idx[s_]] and s_ not in done:
                        done.append(s_)
                g += n
            idle = []
            for s_ in done:
                finished.append((slot_idx[s_], torch.stack(slot_frames[s_][:limits[slot_idx[s_]]]).to(torch.int32)))
                if not start(s_):
                    idle.append(s_)
            if idle and any(i is not None for i in slot_idx):
                m.reset_slots(idle)                                      # a retired slot keeps stepping: keep its position away from max_seq
            yield from finished
            finished.clear()

    def _iter_codes_refilling_beside_the_loop(self, prompts, limits: List[int], temperature: float, topk: int, poll: int, B: int):
        """The continuously refilled batch WITHOUT stalls (round 4): a retired slot's next prompt runs a few backbone layers after
        each frame step (Model.refill_begin / refill_advance: about ``refill_row_layers`` = 600 prompt-row x layer units per step, i.e.
        3 layers of a 190-row prompt = +8 % of a B = 32 step; measured: bench.py extras.config3.refill_beside_the_loop) while the other slots keep generating, and the new utterance's frame
        0 is sampled by the batch's next frame step -- csm_prefill_slot made the other slots wait ~4 ms for a 190-row prompt and
        > 8 ms for a 1,334-row one.  Until its prompt is complete a slot's rows are placeholders and are skipped here."""
        from collections import deque
        m = self._model
        L = getattr(m.bb, "num_layers", 16)
        budget = getattr(self, "refill_row_layers", 600)
        pending = deque(range(len(prompts)))
        free = deque(range(B))
        slot_idx: List[Optional[int]] = [None] * B          # prompt index generating in the slot
        start_g: List[int] = [0] * B                        # global frame index of its frame 0
        slot_frames: List[List[torch.Tensor]] = [[] for _ in range(B)]
        refilling: Optional[Tuple[int, int, int]] = None    # (slot, prompt index, prompt rows)
        m.reset_caches()

        def feed(everything: bool) -> None:
            """One bounded piece of refill work (``everything``: nobody is generating, so run whole prompts)."""
            nonlocal refilling
            while True:
                if refilling is None:
                    if not (free and pending):
                        return
                    slot, i = free.popleft(), pending.popleft()
                    t, mk = prompts[i]
                    m.refill_begin(slot, t, mk)
                    refilling = (slot, i, int(t.shape[0]))
                slot, i, rows = refilling
                # the per-step budget grows with the backlog: every slot that waits for a prompt is 1/B of the batch's throughput idle, and
                # the refill work is the same whenever it is done -- with nobody waiting the steps stay within ~8 % of an undisturbed one
                per_call = max(1, budget * (1 + len(free)) // max(rows, 1))
                if m.refill_advance(L if everything else min(per_call, L)):
                    slot_idx[slot], start_g[slot], slot_frames[slot] = i, m.num_frames(), []
                    refilling = None
                if not everything:
                    return

        feed(True)                                          # the initial fill: nothing to protect yet
        g = m.num_frames()
        while any(i is not None for i in slot_idx) or refilling is not None or (pending and free):
            if not any(i is not None for i in slot_idx):
                feed(True)                                  # only prompts left: finish them at full speed
                continue
            active = [s_ for s_ in range(B) if slot_idx[s_] is not None]
            n = max(min(poll, min(limits[slot_idx[s_]] - len(slot_frames[s_]) for s_ in active)), 1)
            for _ in range(n):
                m.step(B, temperature, topk)
                feed(False)
            fr, eos = m.read_frames(B, g, n)
            done = []
            for s_ in [s_ for s_ in range(B) if slot_idx[s_] is not None]:      # (a slot may have joined during this block)
                lo = max(start_g[s_] - g, 0)                # rows of this block that belong to the slot's current utterance
                if lo >= n:
                    continue                                # (it joined after this block's last step)
                rows, e = fr[lo:, s_], int(eos[s_])
                ended = e >= start_g[s_]
                if ended:
                    rows = rows[: max(e - (g + lo), 0)]
                slot_frames[s_].extend(rows.unbind(0))
                if ended or len(slot_frames[s_]) >= limits[slot_idx[s_]]:
                    done.append(s_)
            g += n
            for s_ in done:
                fs = slot_frames[s_][:max(limits[slot_idx[s_]], 0)]
                yield slot_idx[s_], (torch.stack(fs).to(torch.int32) if fs else torch.empty(0, 32, dtype=torch.int32))
                slot_idx[s_], slot_frames[s_] = None, []
                free.append(s_)
            if free and (pending or any(i is not None for i in slot_idx)):
                m.reset_slots(list(free))                   # retired slots keep stepping as placeholders: keep their positions away from max_seq

    def generate_codes_continuous(self, prompts: Sequence[Tuple[torch.Tensor, torch.Tensor]], max_generation_len: int,
                                  temperature: float, topk: int, poll: Optional[int] = None) -> List[torch.Tensor]:
        """``iter_codes_continuous`` collected: each prompt's frames [n_i][32] int32 (CPU), in the order of ``prompts``."""
        results: List[torch.Tensor] = [torch.empty(0, 32, dtype=torch.int32) for _ in prompts]
        for i, frames in self.iter_codes_continuous(prompts, max_generation_len, temperature, topk, poll):
            results[i] = frames
        return results

    def generate_many(self, texts: Sequence, speakers: Sequence[int], contexts: Sequence[List[Segment]], max_audio_length_ms=90_000,
                      temperature: float = 0.7, topk: int = 30) -> List[torch.Tensor]:
        """``generate`` for a list of requests through the continuously refilled batch: one audio tensor per request
        (``max_audio_length_ms``: one value, or one per request)."""
        max_generation_len = (int(max_audio_length_ms / FRAME_MS) if isinstance(max_audio_length_ms, (int, float))
                              else [int(x / FRAME_MS) for x in max_audio_length_ms])
        prompts = [self._build_prompt(t, sp, ctx) for t, sp, ctx in zip(texts, speakers, contexts)]
        out: List[torch.Tensor] = [torch.tensor([]) for _ in prompts]
        for i, frames in self.iter_codes_continuous(prompts, max_generation_len, temperature, topk):
            if frames.shape[0]:                                       # decoded as each utterance finishes, not at the end
                out[i] = self._decode_frames(frames.unsqueeze(1))
        return out

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

        gate = _FirstBlockGate()
        blocks = self._frame_blocks(tokens, mask, max_generation_len, temperature, topk, size, gate=gate)
        while True:
            with torch.inference_mode():
                fr = next(blocks, None)
            if fr is None:
                break
            pending.extend(fr.unbind(0))
            while len(pending) >= size:
                chunk = decode(size)
                gate.release()                   # the first chunk is decoded: queue the next block before the user gets this one
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


def _wav_float32_header(sample_rate: int, data_bytes: int) -> bytes:
    """44-byte RIFF header of a mono 32-bit IEEE-float WAV (format tag 3) -- what ``torchaudio.save(file, audio.unsqueeze(0), sr)``
    writes for a float32 tensor (reference: generator.py:327); torchaudio is not a dependency here."""
    return (b"RIFF" + struct.pack("<I", 36 + data_bytes) + b"WAVE"
            + b"fmt " + struct.pack("<IHHIIHH", 16, 3, 1, sample_rate, sample_rate * 4, 4, 32)
            + b"data" + struct.pack("<I", data_bytes))


def _float32_bytes(audio: torch.Tensor) -> bytes:
    return audio.detach().to(torch.float32).reshape(-1).cpu().contiguous().numpy().astype("<f4", copy=False).tobytes()


def save_wav_float32(filename: str, audio: torch.Tensor, sample_rate: int) -> None:
    """(n,) float samples -> mono float32 WAV in one go."""
    pcm = _float32_bytes(audio)
    with open(filename, "wb") as f:
        f.write(_wav_float32_header(sample_rate, len(pcm)) + pcm)


class AudioStreamWriter:
    """The reference's streaming file writer (sesameai/generator.py:303-327: ``add_chunk`` from the generation loop, one
    ``write_file`` at the end) with a bounded footprint: every chunk is appended to the OPEN file as it arrives and
    ``write_file`` only patches the two RIFF size fields and closes -- a 90 s utterance never sits in memory as a list of
    tensors, and a run that dies half-way leaves the audio produced so far on disk.  No chunk, no file (the reference
    returns before ``torchaudio.save`` then).  ``add_chunk`` may be called from another thread than ``write_file``.

    Until ``write_file`` the header carries the streaming placeholder 0xFFFFFFFF in both size fields ("length unknown": players
    read to the end of the file), so a writer that is abandoned -- dropped, garbage-collected, the process killed -- leaves a WAV
    that still plays; ``close()`` / ``with`` / ``__del__`` finalise it like ``write_file``.  The reference's public ``audio_chunks``
    list is kept as an attribute for callers that look at it, but stays EMPTY here (the chunks are in the file); ``chunks_written``
    counts them."""

    _UNKNOWN = 0xFFFFFFFF

    def __init__(self, filename, sample_rate):
        self.filename = filename
        self.sample_rate = sample_rate
        self.lock = threading.Lock()
        self.audio_chunks: List[torch.Tensor] = []       # reference attribute; not filled (see the class docstring)
        self.chunks_written = 0
        self._file = None
        self._data_bytes = 0

    def add_chunk(self, chunk):
        pcm = _float32_bytes(chunk)                      # (the device -> host copy happens outside the lock)
        with self.lock:
            if self._file is None:
                self._file = open(self.filename, "wb")
                hdr = bytearray(_wav_float32_header(self.sample_rate, 0))
                hdr[4:8] = struct.pack("<I", self._UNKNOWN); hdr[40:44] = struct.pack("<I", self._UNKNOWN)
                self._file.write(bytes(hdr))
            self._file.write(pcm)
            self._data_bytes += len(pcm)
            self.chunks_written += 1

    def write_file(self):
        with self.lock:
            if self._file is None:
                return
            f, self._file = self._file, None
            f.seek(4); f.write(struct.pack("<I", 36 + self._data_bytes))
            f.seek(40); f.write(struct.pack("<I", self._data_bytes))
            f.close()

    close = write_file

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.write_file()
        return False

    def __del__(self):
        try:
            self.write_file()
        except Exception:
            pass


class _ChunkPlayer:
    """Real-time playback of streamed chunks in arrival order (reference: the ``audio_player`` thread of
    generate_streaming_audio, generator.py:384-404) through the optional ``sounddevice`` package.  One worker drains a queue
    until it meets the end marker ``close()`` posts, so nothing polls with time-outs and every queued chunk is played before
    ``close()`` returns.  Raises ImportError at construction when ``sounddevice`` is missing."""

    _END = object()

    def __init__(self, sample_rate: int):
        import sounddevice
        self._sd, self._rate = sounddevice, sample_rate
        self._chunks: "queue.Queue" = queue.Queue()
        self._worker = threading.Thread(target=self._drain, name="csm-chunk-player")
        self._worker.start()

    def _drain(self) -> None:
        for chunk in iter(self._chunks.get, self._END):
            self._sd.play(chunk.detach().to(torch.float32).cpu().numpy(), self._rate)
            self._sd.wait()

    def submit(self, chunk: torch.Tensor) -> None:
        self._chunks.put(chunk)

    def close(self) -> None:
        self._chunks.put(self._END)
        self._worker.join()


def generate_streaming_audio(generator: "Generator", text, speaker: int, context: List[Segment], output_file: str,
                             max_audio_length_ms: float = 90_000, temperature: float = 0.7, topk: int = 30,
                             play_audio: bool = False):
    """reference: sesameai/generator.py:349-434 (same arguments, same console messages): ``generate_stream`` with every chunk
    appended to ``output_file`` as it is produced and, with ``play_audio``, played as it arrives.  Without ``sounddevice``
    playback is switched off with the reference's message and the file is still written."""
    sinks: List[Callable[[torch.Tensor], None]] = []
    writer = AudioStreamWriter(output_file, generator.sample_rate)
    sinks.append(writer.add_chunk)
    player: Optional[_ChunkPlayer] = None
    if play_audio:
        try:
            player = _ChunkPlayer(generator.sample_rate)
            sinks.append(player.submit)
        except ImportError:
            print("sounddevice library not found. Install with 'pip install sounddevice' to enable real-time playback.")

    print("Generating audio in streaming mode...")
    t0 = time.time()
    try:
        stream = generator.generate_stream(text=text, speaker=speaker, context=context, max_audio_length_ms=max_audio_length_ms,
                                           temperature=temperature, topk=topk)
        for n, chunk in enumerate(stream, start=1):
            for sink in sinks:
                sink(chunk)
            print(f"Generated chunk {n}")
    finally:
        writer.write_file()                              # whatever was generated is a valid WAV, also after an error
        if player is not None:
            player.close()
    print(f"Audio generation completed in {time.time() - t0:.2f} seconds")


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
    gen = Generator(model, audio_tokenizer=mimi, max_batch_size=max_batch_size)
    if os.environ.get("CSM_NO_WARMUP") != "1":
        gen.warm_up()
    return gen
