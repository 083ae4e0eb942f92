"""ORACLE -- test infrastructure only.

CPU restatement of the hot path of jianfch/stable-ts.  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import anything from here, and only as the checker.
The product package (``stable_ts_amd``) never imports it.

Layout
  oracle/whisper/      stand-in for the un-vendored dependency openai-whisper==20250625 (audio, model,
                       decoding, timing, tokenizer) -- importable as ``whisper`` by the reference's own
                       glue when /root/reference is present (tests/golden/make_golden.py)
  oracle/stable.py     restatement of the reference's own glue for this path (decode.py, timing.py,
                       alignment.py:396-429) so the oracle runs where /root/reference does not exist
  oracle/dtw.c         plain-C dtw_cpu+backtrace
"""
