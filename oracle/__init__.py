"""ORACLE -- test infrastructure only.

CPU restatement of the hot path of jianfch/stable-ts.  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import anything from here, and only as the checker.
The product package (``stable_ts_amd``) never imports it.

Pinning status: the reference holds no golden vectors or known-answer tests for this path and the dependency whose
algorithm is restated here (openai-whisper==20250625) is absent offline, so against upstream itself the arithmetic is
"parity unpinned".  What pins it instead (tests/test_oracle_pinning.py, tests/golden/): the independent ports shipped
with ``transformers`` (log-mel, encoder / decoder outputs through the reference's own weight-name map, median filter,
DTW incl. the known-answer vector of SURVEY.md 8c, the timestamp / blank / token suppression rules of the decode loop), and the reference's OWN glue (decode.py, timing.py, transcribe,
align, Aligner, Refiner, locate, WhisperResult.regroup) imported from /root/reference and run on top of this package.

Layout
  oracle/whisper/      stand-in for the un-vendored dependency openai-whisper==20250625 (audio, model,
                       decoding, timing, tokenizer) -- importable as ``whisper`` by the reference's own
                       glue when /root/reference is present (tests/golden/make_golden.py)
  oracle/stable.py     restatement of the reference's own glue for this path (decode.py, timing.py,
                       alignment.py:396-429) so the oracle runs where /root/reference does not exist
  oracle/dtw.c         plain-C dtw_cpu+backtrace
"""
