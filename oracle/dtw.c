/* ORACLE (test infrastructure, never shipped, never on the product path).
 *
 * Plain-C restatement of openai-whisper==20250625 whisper/timing.py::dtw_cpu + backtrace
 * (the numba-jitted CPU path that the reference reaches through stable_whisper/timing.py:195).
 * Recurrence and tie-break as documented in SURVEY.md section 3.4:
 *   cost, trace are float32 (N+1)x(M+1); cost[0][0]=0, else +inf; column-major sweep (j outer);
 *   c0=diag, c1=up, c2=left; strict '<' picks diag / up, every tie falls to 'left' (2);
 *   cost[i][j] = (float)( (double)x[i-1][j-1] + (double)c )   -- f64 add, f32 store.
 * backtrace: trace[0][:]=2, trace[:][0]=1, walk from (N,M) to (0,0).
 * Build: see oracle/Makefile.  Used by oracle/whisper/timing.py and as bench.py's cpu_baseline leg.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

/* x: float64 [N][M] row-major.  Writes the path (text index, time index) in forward order,
 * returns its length (between max(N,M) and N+M-1), or -1 on an internal error. */
int oracle_dtw(const double *x, int N, int M, int32_t *text_idx, int32_t *time_idx)
{
    const size_t W = (size_t)M + 1;
    float *cost = (float *)malloc(sizeof(float) * (size_t)(N + 1) * W);
    signed char *trace = (signed char *)malloc((size_t)(N + 1) * W);
    if (!cost || !trace) { free(cost); free(trace); return -1; }
    for (size_t k = 0; k < (size_t)(N + 1) * W; ++k) { cost[k] = INFINITY; trace[k] = -1; }
    cost[0] = 0.0f;
    for (int j = 1; j <= M; ++j) {
        for (int i = 1; i <= N; ++i) {
            const float c0 = cost[(size_t)(i - 1) * W + (j - 1)];
            const float c1 = cost[(size_t)(i - 1) * W + j];
            const float c2 = cost[(size_t)i * W + (j - 1)];
            float c; signed char t;
            if (c0 < c1 && c0 < c2) { c = c0; t = 0; }
            else if (c1 < c0 && c1 < c2) { c = c1; t = 1; }
            else { c = c2; t = 2; }
            cost[(size_t)i * W + j] = (float)(x[(size_t)(i - 1) * M + (j - 1)] + (double)c);
            trace[(size_t)i * W + j] = t;
        }
    }
    for (int j = 0; j <= M; ++j) trace[j] = 2;
    for (int i = 0; i <= N; ++i) trace[(size_t)i * W] = 1;
    int i = N, j = M, n = 0;
    while (i > 0 || j > 0) {
        text_idx[n] = i - 1; time_idx[n] = j - 1; ++n;
        const signed char t = trace[(size_t)i * W + j];
        if (t == 0) { --i; --j; }
        else if (t == 1) { --i; }
        else if (t == 2) { --j; }
        else { free(cost); free(trace); return -1; }
    }
    for (int a = 0, b = n - 1; a < b; ++a, --b) {
        int32_t s = text_idx[a]; text_idx[a] = text_idx[b]; text_idx[b] = s;
        s = time_idx[a]; time_idx[a] = time_idx[b]; time_idx[b] = s;
    }
    free(cost); free(trace);
    return n;
}
