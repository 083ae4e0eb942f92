#!/bin/bash
# round 6 call 37: the scoring pass enqueued from a helper thread while the caller splits the words (engine.score_start; call 36: the
# enqueue keeps the calling thread inside the library for 14 ms of a 19-ms word-timestamp stage): GPU tests that run transcribe() / align()
# end to end, then two processes each, alternating, with and without (--no-score-thread): headline, align(), span mode
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_gpu_golden.py tests/test_gpu_batch_invariance.py tests/test_gpu_f16_bench_windows.py tests/test_gpu_largev3.py -m gpu -q -x 2>&1 | tail -3 ) > gpurun_out/r06_c37_tests.log; cat gpurun_out/r06_c37_tests.log
for rep in 1 2; do
for v in thread nothread; do
    X=""; [ $v = nothread ] && X="--no-score-thread"
    ( timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-f32 --no-roofline $X ) >> gpurun_out/r06_c37_bench_$v.txt 2>> gpurun_out/r06_c37.err
    ( timeout 600 python bench.py --mode align --steps 2 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline $X ) >> gpurun_out/r06_c37_align_$v.txt 2>> gpurun_out/r06_c37.err
    ( timeout 600 python bench.py --spans 20 --steps 2 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline $X ) >> gpurun_out/r06_c37_spans_$v.txt 2>> gpurun_out/r06_c37.err
done
done
( timeout 600 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-f32 --no-roofline --phase-times > gpurun_out/r06_c37_bench_phase.json 2>> gpurun_out/r06_c37.err )
python - <<'PY'
import json
for m in ("bench", "align", "spans"):
    for v in ("thread", "nothread"):
        rows = [json.loads(x) for x in open(f"gpurun_out/r06_c37_{m}_{v}.txt") if x.startswith("{")]
        print(m, v, [r["ms_per_step"] for r in rows], [r["config"].get("words") for r in rows])
d = json.loads(open("gpurun_out/r06_c37_bench_phase.json").read().strip().splitlines()[-1])
print(json.dumps(d.get("phase_ms"), indent=0))
PY
tail -3 gpurun_out/r06_c37.err
