"""One table out of the per-case report of tests/test_gpu_f16_bench_windows.py (gpurun_out/f16_bench_windows_report.json or a copy under
profiles/): per window and mode -- greedy identical?, beam-5 winner identical?, DTW path identical?, words within 20 ms, per-token |d log p|.
usage: python scripts/summarize_bench_windows.py profiles/r06_bench_windows_all20_report.json [summary.json]"""
import json
import sys


def main():
    rep = json.load(open(sys.argv[1]))
    wins = sorted({int(k.split("/")[0][6:]) for k in rep})
    out = {"windows": wins, "modes": {}}
    for dt in ("f32", "f16"):
        rows = []
        for k in wins:
            g, b, w = (rep.get(f"window{k}/{c}112/{dt}") for c in ("greedy", "beam", "words"))
            rows.append(dict(
                window=k,
                greedy_identical=g and g["identical"], greedy_identical_prefix=g and g["identical_prefix"],
                greedy_d_avg_logprob=g and g.get("d_avg_logprob", g.get("d_avg_logprob_same_sequence")),
                greedy_gap_at_divergence=g and g.get("gap_at_first_divergence"),
                beam_winner_identical=b and b["winner_identical"], beam_final_sets_identical=b and b.get("final_beam_sets_identical"),
                beam_d_avg_logprob=b and b.get("d_avg_logprob", b.get("d_avg_logprob_same_sequence")),
                beam_first_diverging_step=b and b.get("first_diverging_step"),
                dtw_path_identical=w and w["dtw_path_identical"], detours=w and len(w["detours"]),
                max_detour_extra_cost_rel=w and max([d["extra_cost_rel"] for d in w["detours"]], default=0.0),
                words=w and w["words"], within_20ms=w and w["within_20ms"], max_dt=w and w["max_dt"],
                max_abs_dlogp=w and w["max_abs_dlogp"], mean_abs_dlogp=w and w["mean_abs_dlogp"]))
        n = len(rows)
        cnt = lambda key: sum(1 for r in rows if r[key])
        mx = lambda key: max((r[key] for r in rows if r[key] is not None), default=None)
        out["modes"][dt] = dict(
            rows=rows, n_windows=n, greedy_identical=cnt("greedy_identical"), beam_winner_identical=cnt("beam_winner_identical"),
            dtw_path_identical=cnt("dtw_path_identical"), all_words_within_20ms=sum(1 for r in rows if r["within_20ms"] == 1.0),
            min_within_20ms=min((r["within_20ms"] for r in rows if r["within_20ms"] is not None), default=None),
            max_greedy_d_avg_logprob=mx("greedy_d_avg_logprob"), max_beam_d_avg_logprob=mx("beam_d_avg_logprob"),
            max_abs_dlogp_per_token=mx("max_abs_dlogp"), words_total=sum(r["words"] or 0 for r in rows))
        s = out["modes"][dt]
        print(f"{dt}: {n} windows | greedy identical {s['greedy_identical']} | beam-5 winner identical {s['beam_winner_identical']} | "
              f"DTW path identical {s['dtw_path_identical']} | all words within 20 ms {s['all_words_within_20ms']} (min share {s['min_within_20ms']}) | "
              f"max |d avg logprob| greedy {s['max_greedy_d_avg_logprob']} beam {s['max_beam_d_avg_logprob']} | max per-token |d log p| {s['max_abs_dlogp_per_token']}")
        for r in rows:
            if not (r["greedy_identical"] and r["beam_winner_identical"] and r["dtw_path_identical"]):
                print("   ", {k: v for k, v in r.items() if k in ("window", "greedy_identical", "greedy_identical_prefix", "greedy_gap_at_divergence",
                                                                  "beam_winner_identical", "beam_first_diverging_step", "dtw_path_identical",
                                                                  "detours", "max_detour_extra_cost_rel", "within_20ms", "max_dt")})
    if len(sys.argv) > 2:
        json.dump(out, open(sys.argv[2], "w"), indent=1)


if __name__ == "__main__":
    main()
