"""``locate()``: find when given words are spoken without transcribing the whole recording.

Behavioural contract = ``stable_whisper/alignment.py::locate`` (:756-1116).  Per 30-s chunk:

1. one teacher-forced pass over ``initial_tokens + text_tokens`` with the alignment heads' cross-attention captured; the
   frame where the LAST text token attends most (after softmax over frames, z-normalisation over tokens, median-7 and
   the head mean -- the same processing as the word-timestamp path, but over every token row) approximates where the
   text ends (:930-951);
2. (modes 0, 1) the ``duration_window`` seconds around that time are decoded greedily, token by token, while the search
   text is forced in whenever it is likely enough (probability, exact arg-max, or string match) (:975-1065);
3. (mode 0) the decoded tokens get word timestamps against the whole chunk (:1089-1107).

Device work goes through four engine calls that exist for the hot path anyway: log-mel + encoder + cross-KV, the scoring
pass (``engine.score`` with no row cropping: ``n_sot = 0``), and ``engine.forward_logits``.  The greedy loop is host
logic with one logits call per token (<= ``max_token_per_seg`` + 1 tokens).  It mirrors the reference's bookkeeping
exactly, including what its KV cache holds after a partly matched text is rolled back (:1040-1048: the cache is cleared
and decoding continues from the single next token) -- observable behaviour, so it is kept.  Checked against the
reference's ``locate`` on the CPU oracle through the engine stand-in (tests/test_locate_cpu.py).
"""
from typing import List, Optional, Tuple, Union

import numpy as np
import torch

from .stabilization import host_single_thread
from .audio import FRAMES_PER_SECOND, N_FFT, N_FRAMES, N_SAMPLES, SAMPLE_RATE
from .decoding import DecodingOptions, DecodingPlan
from .result import Segment
from .timing import add_word_timestamps_batch, split_word_tokens

CHUNK_LENGTH = N_SAMPLES // SAMPLE_RATE


def _chunk_mel(model, audio_segment: torch.Tensor) -> torch.Tensor:
    """log-mel [n_mels, 3000] of one chunk as the reference computes it for locate (:924-925): the segment plus
    N_FFT // 2 + 1 zero samples, trimmed / zero-filled to 3000 frames (swx_log_mel_ragged)."""
    return model.log_mel_segments([audio_segment], padding=N_FFT // 2 + 1)[0]


def _pad_frames(mel: torch.Tensor) -> torch.Tensor:
    out = torch.zeros(mel.shape[0], N_FRAMES, dtype=mel.dtype, device=mel.device)
    k = min(N_FRAMES, mel.shape[-1])
    out[:, :k] = mel[:, :k]
    return out


@host_single_thread
def locate(model, audio, text: Union[str, List[int]], language: str, count: int = 1,
           duration_window: Union[float, Tuple[float, float]] = 3.0, *, mode: int = 0, start: float = None,
           end: float = None, probability_threshold: float = 0.5, eots: int = 1, max_token_per_seg: int = 20,
           exact_token: bool = False, case_sensitive: bool = False, verbose: Optional[bool] = False,
           initial_prompt: str = None, suppress_tokens: Union[str, List[int]] = "-1", **unsupported):
    from .transcribe import as_waveform, pop_audio_options
    audio_options = pop_audio_options(unsupported)
    if unsupported:
        raise TypeError(f"locate() got unexpected keyword argument(s): {sorted(unsupported)}")
    sec_per_emb = model.dims.n_audio_ctx / CHUNK_LENGTH
    if isinstance(duration_window, (float, int)):
        duration_window = [duration_window] * 2
    assert N_SAMPLES > sum(duration_window), f"Sum of [duration_window] must be less than {N_SAMPLES}, got {sum(duration_window)}"
    adjusted_chunk = N_SAMPLES - round(duration_window[0] * SAMPLE_RATE)
    if initial_prompt:
        initial_prompt = " " + initial_prompt.strip()
    plan = DecodingPlan(model, DecodingOptions(language=language, prompt=initial_prompt, suppress_tokens=suppress_tokens,
                                               without_timestamps=True))
    tok = plan.tokenizer
    initial_tokens = list(plan.initial_tokens)
    text_tokens, text = (tok.encode(text), text) if isinstance(text, str) else (list(text), tok.decode(list(text)))
    if not exact_token and not case_sensitive:
        text = text.lower()
    suppressed = [t for t in plan.suppress if t < tok.eot]
    eng = model.engine

    audio = as_waveform(audio, **audio_options).detach().float().cpu()
    if end:
        audio = audio[:round(end * SAMPLE_RATE)]
    seek_sample = round(start * SAMPLE_RATE) if start else 0
    total = int(audio.shape[-1])
    found = 0
    prev_target_end = None
    matches = []

    def step_logits(xkv, seq: List[int]) -> torch.Tensor:
        return eng.forward_logits(xkv, [seq])[0, len(seq) - 1, : tok.eot + 1].float().cpu().clone()

    while seek_sample < total and (not count or found < count):
        seek = round(seek_sample / SAMPLE_RATE, 3)
        chunk = audio[seek_sample: seek_sample + N_SAMPLES]
        mel = _chunk_mel(model, chunk)
        xkv = model.cross_kv(model.encoder(mel[None]))
        # -- 1. where does the text end?  alignment matrix over every row of initial_tokens + text_tokens
        seq = initial_tokens + text_tokens
        _, neg, _ = eng.score(xkv, [seq + [tok.eot]], [model.dims.n_audio_ctx], n_sot=0, eot=tok.eot)
        last_row = (-neg[0, len(seq) - 1, : model.dims.n_audio_ctx]).float().cpu()
        target_end = round((last_row.argmax() / sec_per_emb).item(), 3)
        if mode == 2:
            found += 1
            if seek_sample + N_SAMPLES >= total or (count and found >= count) or prev_target_end == target_end:
                seek_sample = total
            else:
                seek_sample += round(target_end * SAMPLE_RATE)
            prev_target_end = target_end
            matches.append(dict(tokens=[], target_end=target_end + seek))
            continue
        # -- 2. greedy decode of the duration window with the search text forced in
        curr_start = round(max(target_end - duration_window[0], 0.0), 3)
        curr_end = round(target_end + duration_window[1], 3)
        section = _pad_frames(mel[..., round(curr_start * FRAMES_PER_SECOND): round(curr_end * FRAMES_PER_SECOND)])
        xkv_sec = model.cross_kv(model.encoder(section[None]))
        cache: List[int] = []                    # what the reference's KV cache holds = the context of the next logits
        feed: List[int] = list(initial_tokens)   # `temp_tokens`: what is appended to the cache by the next call
        fed_log: List[List[int]] = [list(initial_tokens)]      # `infer_tokens`
        predictions = []
        target_idx = 0
        running, found_target = True, False
        curr_eots = 0
        to_decode: List[int] = []
        replaced: List[int] = []
        while running:
            cache = cache + feed
            logits = step_logits(xkv_sec, cache)
            logits[suppressed] = -np.inf
            top2 = logits.sort(dim=-1).indices[-2:]
            best = int(top2[-1])
            best_non_eot = int(top2[-2]) if best == tok.eot else best
            probs = logits[: tok.eot].softmax(dim=-1)
            # The reference keeps `best_token` / `best_non_eot_token` as views of one small tensor: when the arg-max is
            # not EOT they are the SAME element, so forcing the target token in place (:1022) also rewrites the copy that
            # was just queued for string matching and the value compared two lines later.  `same_slot` carries that.
            same_slot = best != tok.eot
            queued = False
            if found_target:
                target_prob = is_match = None
            else:
                if exact_token:
                    is_match = False
                else:
                    to_decode.append(best_non_eot)
                    queued = True
                    temp_text = tok.decode(to_decode)
                    if not case_sensitive:
                        temp_text = temp_text.lower()
                    is_match = temp_text.endswith(text)
                    if is_match:
                        to_decode = []
                        queued = False
                target_prob = probs[text_tokens[target_idx]].item()
            if target_prob is not None and (target_prob >= probability_threshold or
                                            best_non_eot == text_tokens[target_idx] or is_match):
                if is_match:
                    best = best_non_eot
                    token_prob = probs[best].item()
                    found_target = True
                else:
                    best = text_tokens[target_idx]
                    if same_slot:
                        best_non_eot = best
                        if queued:
                            to_decode[-1] = best
                    if replaced or best_non_eot != text_tokens[target_idx]:
                        replaced.append(best_non_eot)
                    target_idx += 1
                    if target_idx == len(text_tokens):
                        found_target = True
                    token_prob = target_prob
                if found_target:
                    found += 1
                curr_eots = 0
            else:
                if not found_target:
                    if replaced:
                        # :1040-1046 rebuilds the decoder input with torch.cat of a [1, n - k] and a [1, k] tensor along
                        # dim 0, which only works for n - k == k; the same call is made here so that the same error
                        # surfaces.  The rebuilt input is then overwritten below; what remains is the cleared cache.
                        n_fed = sum(len(x) for x in fed_log)
                        torch.cat([torch.zeros(1, n_fed - len(replaced), dtype=torch.long),
                                   torch.zeros(1, len(replaced), dtype=torch.long)])
                        replaced = []
                        cache = []
                    target_idx = 0
                if best == tok.eot:
                    if curr_eots >= eots or found_target:
                        running = False
                    else:
                        curr_eots += 1
                        best = best_non_eot
                else:
                    curr_eots = 0
                token_prob = None if best == tok.eot else probs[best].item()
            predictions.append(dict(token=best, prob=token_prob))
            if len(predictions) > max_token_per_seg:
                running = False
            if running:
                fed_log.append([best])
                feed = [best]
        match = None
        if found_target:
            final_tokens = [p["token"] for p in predictions]
            if mode == 1:
                _, (ws, wts), _ = split_word_tokens([dict(tokens=final_tokens)], tok)
                tprobs = [p["prob"] for p in predictions]
                wps = [float(np.mean([tprobs.pop(0) for _ in wt])) for wt in wts]
                words = [dict(word=w, tokens=wt, probability=wp) for w, wt, wp in zip(ws, wts, wps)]
                match = dict(end=target_end + seek, text=text, duration_window_text="".join(ws), duration_window_word=words)
                seek_sample += round(curr_end * SAMPLE_RATE)
            else:
                seg = dict(seek=0, tokens=final_tokens)
                add_word_timestamps_batch(model=model, tokenizer=tok, xkv=xkv, gap_padding=None,
                                          windows=[dict(segments=[seg], num_samples=round(curr_end * SAMPLE_RATE))])
                match = Segment(words=seg["words"])
                seek_sample += round(match.words[-1].end * SAMPLE_RATE)
                match.offset_time(seek)
                match.seek = curr_start
            if verbose:
                print(f'Confirmed: "{text}" ending at ~{target_end + seek:.3f}s')
        else:
            seek_sample += adjusted_chunk if chunk.shape[-1] == N_SAMPLES else int(chunk.shape[-1])
        if match:
            matches.append(match)
    if verbose and not matches:
        print(f'Failed to locate "{text}".')
    return matches
