"""Default regrouping hook of transcribe()/align() (result.py:2893-3024, default algorithm string at :3008).

The regroup DSL (split by punctuation / gap / length, merge, clamp) is pure list surgery on ``Segment.words`` that runs
after the hot path; SURVEY.md 8f ranks it "next-1".  This round keeps the segments exactly as decoded (word start/end
are unaffected by every regroup step except ``cm``), so timestamps stay comparable with the oracle.
"""


def regroup_default(result, regroup=True):
    return result
