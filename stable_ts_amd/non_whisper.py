"""``transcribe_any``: stable-ts post-processing around ANY speech recogniser (seam B4).

Mirrors ``stable_whisper/non_whisper/transcribe.py:26-370``: the audio is converted to the kind of object the caller's
``inference_func`` wants (path, file bytes, tensor or array, at the model's sample rate), the function's output -- a
``WhisperResult`` or anything ``WhisperResult`` accepts (dict, list of segment dicts, list of word-dict lists) -- gets
the loudness-based silence suppression and the default regrouping.  Nothing here touches the GPU path; it is the seam
through which the reference's other back ends (faster-whisper, HF, MLX) reuse the result model, kept so that code
written against it keeps working.  Denoisers and the Silero VAD are out of scope.
"""
import io
import os
import warnings
import wave
from typing import Callable, Optional, Union

import numpy as np
import torch

from .stabilization import _single_host_thread
from .audio_io import (AudioLoader, audio_to_tensor_resample, check_source, get_samplerate, load_audio, reject_denoiser,
                       resample, to_s16, voice_freq_filter, write_wav)
from .result import WhisperResult

AUDIO_TYPES = ("str", "byte", "torch", "numpy")
# what the type of ``audio`` implies when ``audio_type`` is not given; as in the reference the implied names for tensors
# and bytes are not members of AUDIO_TYPES, so such inputs are handed to ``inference_func`` as they are
AUDIO_TYPE_BY_CLASS = {str: "str", bytes: "bytes", np.ndarray: "numpy", torch.Tensor: "pytorch", AudioLoader: None}


def _wav_bytes(audio: torch.Tensor, sr: int) -> bytes:
    with io.BytesIO() as f:
        a = audio.detach().cpu().numpy()
        a = a[None] if a.ndim == 1 else a
        with wave.open(f, "wb") as w:
            w.setnchannels(a.shape[0])
            w.setsampwidth(2)
            w.setframerate(sr)
            w.writeframes(to_s16(a.T.reshape(-1)).tobytes())
        return f.getvalue()


def transcribe_any(inference_func: Callable, audio: Union[str, np.ndarray, torch.Tensor, bytes, AudioLoader],
                   audio_type: Optional[str] = None, input_sr: Optional[int] = None, model_sr: Optional[int] = None,
                   inference_kwargs: Optional[dict] = None, temp_file: Optional[str] = None, verbose: Optional[bool] = False,
                   regroup: Union[bool, str] = True, suppress_silence: bool = True, suppress_word_ts: bool = True,
                   q_levels: int = 20, k_size: int = 5, denoiser: Optional[str] = None, denoiser_options: Optional[dict] = None,
                   demucs: bool = False, demucs_options: Optional[dict] = None, vad: Union[bool, dict] = False,
                   vad_threshold: float = 0.35, vad_onnx: bool = False, min_word_dur: Optional[float] = None,
                   min_silence_dur: Optional[float] = None, nonspeech_error: float = 0.1, use_word_position: bool = True,
                   only_voice_freq: bool = False, only_ffmpeg: bool = False, force_order: bool = False,
                   check_sorted: bool = True) -> WhisperResult:
    reject_denoiser(denoiser, demucs)
    if vad:
        raise NotImplementedError("vad needs the Silero model (torch.hub, network) -- out of scope offline")
    if audio_type is not None and (audio_type := audio_type.lower()) not in AUDIO_TYPES:
        raise NotImplementedError(f'``audio_type="{audio_type}"`` is not supported. Types: {AUDIO_TYPES}')
    if isinstance(audio, AudioLoader) and audio_type is not None:
        raise ValueError(f"``audio_type`` can only be ``None`` when ``audio`` is an AudioLoader instance,but got {audio_type}")
    if audio_type is None:
        if type(audio) not in AUDIO_TYPE_BY_CLASS:
            raise TypeError(f"{type(audio)} is not supported for ``audio``.")
        audio_type = AUDIO_TYPE_BY_CLASS[type(audio)]
    is_array = isinstance(audio, (np.ndarray, torch.Tensor))
    if input_sr is None and is_array and (only_voice_freq or suppress_silence or model_sr):
        raise ValueError("``input_sr`` is required when ``audio`` is a PyTorch tensor or NumPy array.")
    if model_sr is None and isinstance(audio, (str, bytes)) and audio_type in ("torch", "numpy"):
        raise ValueError('``model_sr`` is required when ``audio_type`` is a "pytorch" or "numpy".')
    if isinstance(audio, str):
        check_source(audio)
    inference_kwargs = {} if inference_kwargs is None else inference_kwargs
    temp_file = os.path.abspath(temp_file or "./_temp_stable-ts_audio_.wav")
    temp_audio_file = None

    if isinstance(audio, AudioLoader):
        if only_voice_freq and not audio._only_voice_freq:
            warnings.warn("``only_voice_freq=True`` will have no affect unless specified at AudioLoader initialization.", stacklevel=2)
        only_voice_freq = False
        if suppress_silence:
            warnings.warn("``suppress_silence=True`` is not yet supported when ``audio`` is an AudioLoader.", stacklevel=2)
        suppress_silence = False
        if input_sr is not None and input_sr != audio.sr:
            warnings.warn(f"``input_sr`` ({input_sr}) does not match ``sr`` of AudioLoader ({audio.sr})", stacklevel=2)
        input_sr = audio.sr

    encoded = isinstance(audio, (str, bytes))
    audio_sr = input_sr

    def current_sr(optional: bool = False) -> Optional[int]:
        nonlocal audio_sr
        if optional and encoded:
            return None
        if audio_sr is None:
            assert isinstance(audio, (str, bytes)), "No ``input_sr`` specified."
            audio_sr = get_samplerate(audio)
            assert audio_sr is not None, "Failed to get samplerate from ``audio``"
        return audio_sr

    if only_voice_freq:
        if encoded and audio_sr and model_sr:
            audio_sr = max(audio_sr, model_sr)
        audio = audio_to_tensor_resample(audio, original_sample_rate=current_sr(), verbose=verbose, only_ffmpeg=only_ffmpeg)
        audio = voice_freq_filter(audio, audio_sr)
        encoded = False

    final, final_sr = audio, audio_sr
    if model_sr is not None:
        final_sr = current_sr()
        if final_sr != model_sr:
            if isinstance(final, (str, bytes)):
                final, final_sr = load_audio(final, sr=model_sr, verbose=verbose, only_ffmpeg=only_ffmpeg), model_sr
            else:
                if isinstance(final, np.ndarray):
                    final = torch.from_numpy(final)
                if isinstance(final, torch.Tensor):
                    final, final_sr = resample(final, audio_sr, model_sr), model_sr

    if audio_type in ("torch", "numpy"):
        if isinstance(final, (str, bytes)):
            final = load_audio(final, sr=model_sr, verbose=verbose, only_ffmpeg=only_ffmpeg)
        if not isinstance(final, AudioLoader):
            if audio_type == "torch":
                if isinstance(final, np.ndarray):
                    final = torch.from_numpy(final)
            elif isinstance(final, torch.Tensor):
                final = final.cpu().numpy()
    elif audio_type == "str":
        if isinstance(final, (torch.Tensor, np.ndarray)):
            write_wav(temp_file, torch.as_tensor(final), final_sr)
            final = temp_audio_file = temp_file
        elif isinstance(final, bytes):
            with open(temp_file, "wb") as f:
                f.write(final)
            final = temp_audio_file = temp_file
    elif audio_type == "byte":
        if isinstance(final, (torch.Tensor, np.ndarray)):
            final = _wav_bytes(torch.as_tensor(final), final_sr)
        elif isinstance(final, str):
            with open(final, "rb") as f:
                final = f.read()

    inference_kwargs["audio"] = final
    result = None
    try:
        result = inference_func(**inference_kwargs)
        if not isinstance(result, WhisperResult):
            result = WhisperResult(result, force_order=force_order, check_sorted=check_sorted)
        if suppress_silence:
            # this package's own host section (element-wise passes over the waveform): the intra-op pool is parked here and
            # only here -- `inference_func` above ran with the caller's thread settings
            with _single_host_thread():
                result.adjust_by_silence(audio, vad, vad_onnx=vad_onnx, vad_threshold=vad_threshold, q_levels=q_levels,
                                         k_size=k_size, sample_rate=current_sr(True), min_word_dur=min_word_dur,
                                         word_level=suppress_word_ts, verbose=verbose, nonspeech_error=nonspeech_error,
                                         use_word_position=use_word_position, min_silence_dur=min_silence_dur)
            result.set_current_as_orig()
        if result.has_words and regroup:
            result.regroup(regroup)
    finally:
        if temp_audio_file is not None:
            try:
                os.unlink(temp_audio_file)
            except Exception as e:                                                     # noqa: BLE001
                warnings.warn(f"Failed to remove temporary audio file {temp_audio_file}. {e}")
    return result
