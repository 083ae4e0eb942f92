// swx_decode.h -- device-resident bookkeeping of one decode job (W windows x G sequences)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/swx.h"

struct DecodeBufs {
    swx_decode_cfg cfg;
    int W, G, M, V, TS, n_ctx, n_init;
    int max_cand, fin_cap;
    int32_t *tokens[2];        // [M][TS] (double buffered for the beam gather)
    int32_t *anc[2];           // [M][n_ctx] ancestor tables (null when G == 1)
    int32_t *pos0;             // [M] position of the token the next forward pass embeds
    float *sum_lp, *sum_lp_next;   // [M]
    int32_t *row_done;         // [M]
    int32_t *win_done, *win_done_prev;   // [W]
    int32_t *n_done;           // [1]
    int32_t *step_dev;         // [1] number of tokens sampled so far: read by the selection kernels, advanced by the step-finish
                               //     kernel -- the step index is device state so that a captured step graph can be replayed
    int32_t *fin_tokens;       // [W][fin_cap][TS]
    float *fin_score;          // [W][fin_cap]
    int32_t *fin_len;          // [W][fin_cap]
    int32_t *fin_count;        // [W]
    float *cand_lp;            // [M][G+1]
    int32_t *cand_tok;         // [M][G+1]
    float *logits;             // [M][V] f32
    const int32_t *suppress;   // [n_suppress]
    const uint8_t *ts_mask;    // [W][1501] or null
    const int32_t *win_uid;    // [W] stable window identities for the sampling RNG, or null (= window index)
};

int swx_decode_init(const DecodeBufs &b, const int32_t *init_tokens, hipStream_t s);
int swx_decode_after_prefill(const DecodeBufs &b, const float *lg2, float *nospeech, hipStream_t s);
int swx_decode_select(const DecodeBufs &b, int cur, hipStream_t s);
int swx_decode_finalize(const DecodeBufs &b, int cur, int n_steps, int32_t *tokens_out, int32_t *lens_out,
                        float *sumlp_out, int G_out, hipStream_t s);
