"""Hardware check of the un-split ("fat workgroup") decode GEMM variants selected by SWX_PG_POLICY: decode with the
fused step under the policy vs the generic per-op path in the same process (same criterion as
tests/test_gpu_model.py::test_decode_f16_fast_step_equals_general_path).  Exit code 0 = agreement.

    SWX_PG_POLICY="1536x384=1,1152x384=1,384x384=1,384x1536=2" python tests/hw_checks/pg_policy_check.py tiny.en
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main(name: str) -> int:
    import test_gpu_model as T                       # helpers: oracle / engine construction, token config
    from oracle import stable as ost
    from oracle.whisper.decoding import DecodingOptions
    from stable_ts_amd import _lib
    lib = _lib.load()
    m, eng = T._oracle(name), T._engine(name, "f16")
    mels = T._mel(m.dims.n_mels, 71, B=3)
    worst = 1.0
    for beam in (False, True):
        task = ost.DecodingTaskStable(m, DecodingOptions(fp16=False, language="en", max_initial_timestamp=None, sample_len=24,
                                                         beam_size=5 if beam else None))
        kw = dict(n_group=task.n_group, beam=beam, sample_len=24, sot_index=task.sot_index, min_tokens=24,
                  **T._tok_cfg(task.tokenizer, task))
        xkv = eng.cross_kv(eng.encode(mels.cuda().contiguous()))
        fast = eng.decode(xkv, [list(task.initial_tokens)] * 3, **kw)
        old = lib.swx_debug_flags(1)
        try:
            slow = eng.decode(xkv, [list(task.initial_tokens)] * 3, **kw)
        finally:
            lib.swx_debug_flags(old)
        sb = fast["sample_begin"]
        agree = 0
        for w in range(3):
            a = fast["tokens"][w, T._rank(fast, w), sb:sb + 24].tolist()
            b = slow["tokens"][w, T._rank(slow, w), sb:sb + 24].tolist()
            n = 0
            for x, y in zip(a, b):
                if x != y:
                    break
                n += 1
            agree += n
        frac = agree / (3 * 24)
        worst = min(worst, frac)
        ok_nsp = np.allclose(fast["no_speech_prob"], slow["no_speech_prob"], rtol=2e-2, atol=1e-6)
        print(f"{name} beam={beam} policy={os.environ.get('SWX_PG_POLICY')}: common prefix {frac:.2f}, no_speech ok={ok_nsp}")
        if not ok_nsp:
            return 1
    return 0 if worst >= 0.6 else 1


if __name__ == "__main__":
    sys.exit(main(sys.argv[1] if len(sys.argv) > 1 else "tiny.en"))
