"""Seam B3 on hardware: stable_ts_amd.alignment.make_refinement_func vs the golden produced by the reference's
get_whisper_refinement_func on the CPU oracle (tests/golden/make_golden.py::run_refine_case).  Exit code 0 = parity.

    python tests/hw_checks/b3_check.py            (needs a GPU; run by tests/test_gpu_golden.py in a subprocess)
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(os.path.dirname(HERE), "golden")
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, GOLDEN)


def main() -> int:
    import stable_ts_amd as sw
    from make_golden import refine_probe_audio
    from stable_ts_amd.alignment import make_refinement_func
    from stable_ts_amd.tokenizer import get_tokenizer
    with open(os.path.join(GOLDEN, "reference_glue.json")) as f:
        g = json.load(f)["refine_tiny_en"]
    case = g["case"]
    dims = sw.dims_for(case["model"])
    model = sw.Whisper(dims, dtype="f32", max_windows=2, max_rows=5)
    model.load_state_dict(sw.random_state_dict(dims, seed=1234, std=0.02, embed_gain=case["gain"], ts_gain=case["ts_gain"]))
    tok = get_tokenizer(False, num_languages=model.num_languages)
    probs = make_refinement_func(model, tok)(refine_probe_audio(case), g["ids"])
    assert tuple(probs.shape) == (2, len(g["ids"]), tok.eot), tuple(probs.shape)
    pos = torch.arange(len(g["ids"]))
    true_p = probs[:, pos, g["ids"]].float().cpu().numpy()
    want = np.asarray(g["true_prob"])
    worst = float(np.abs(true_p / want - 1).max())
    agree = float((probs.argmax(-1).cpu().numpy() == np.asarray(g["top1"])).mean())
    print(f"true-token probability: worst relative deviation {worst:.3e}; arg-max agreement {agree:.3f}")
    return 0 if (worst <= 2e-2 and agree >= 0.95) else 1


if __name__ == "__main__":
    sys.exit(main())
