/*
 * oracle/viterbi_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C CPU restatement of the reference's Viterbi decode operator
 * (torbi::viterbi_decode, CPU dispatch).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; torbi_amd/ never does.
 *
 * Parity status: PINNED.  tests/test_oracle.py checks this file against
 *   (a) the reference's only known-answer test (tests/test_core.py:7-25 -> [1,2,2]),
 *   (b) committed golden vectors in tests/golden/ that were produced by the
 *       reference's own C++ operator (csrc/ops.cpp + csrc/viterbi.cpp compiled
 *       unchanged into oracle/_ref/, see oracle/build.py and
 *       tests/golden/generate.py), and
 *   (c) when oracle/_ref/ is present, the reference operator itself on fresh
 *       seeded inputs.
 *
 * What is restated (all citations are into /root/reference/torbi/csrc/viterbi.cpp):
 *   forward pass      viterbi_make_trellis_cpu       :35-120
 *   backtrace         viterbi_backtrace_trellis_cpu  :140-160
 *   orchestration     viterbi_decode_cpu             :182-234
 *
 * Result-defining conventions kept bit-for-bit:
 *   - transition is indexed [next, prev]: candidate(j, i) = post[i] + trans[j*S+i]  (:81-86)
 *   - two fp32 roundings per cell, no fused ops: c = post[i] + trans[j,i];
 *     post'[j] = obs[t,j] + max_i c                                              (:84,:102)
 *   - running max starts at i = 0 and is replaced only on a strict '>' so the
 *     lowest index wins ties; the backpointer stays 0 unless replaced because the
 *     trellis is zero-initialised                                                (:94-100,:201-203)
 *   - t = 0 is obs[0,i] + initial[i]                                              (:74)
 *   - final state = first maximal index of the last posterior row (ATen argmax), written
 *     to EVERY column of the item's output row, including t >= frames            (:218-221)
 *   - backtrace from t = frames-1 down to 1                                       (:153-157)
 *
 * Two forward variants with identical results:
 *   mode 0  "reference-shaped": S*S scratch pass (with the reference's integer modulo)
 *           followed by a per-row scan -- the cost structure of :81-104; this is what
 *           bench.py times as cpu_baseline kind "port".
 *   mode 1  "fused": one pass per row, no scratch (faster; used by the tests).
 *
 * Build: gcc -O3 -fopenmp -shared -fPIC (no -ffast-math: the reference is built with
 * -O3 -fopenmp only, setup.py:60-65).  -ffp-contract=off is passed for safety although
 * there are no multiplies to contract.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_OK 0
#define ORACLE_EARG -1
#define ORACLE_ENOMEM -2

int torbi_oracle_abi_version(void) { return 1; }

int torbi_oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* forward pass for ONE batch item; trellis is T*S, zero-initialised by the caller */
static void forward_item(const float *obs, const float *trans, const float *init,
                         float *post_cur, float *post_next, float *scratch,
                         int32_t *trellis, float *post_out, int frames, int S,
                         int mode, int nthreads) {
    const long S2 = (long)S * S;
    (void)nthreads;
    /* viterbi.cpp:72-76 */
    #pragma omp parallel for schedule(static) num_threads(nthreads)
    for (int i = 0; i < S; i++) post_cur[i] = obs[i] + init[i];

    for (int t = 1; t < frames; t++) {
        const float *obs_t = obs + (long)t * S;
        int32_t *tr_t = trellis + (long)t * S;
        if (mode == 0) {
            /* viterbi.cpp:81-86 : probability[i] = posterior_current[i % S] + transition[i] */
            #pragma omp parallel for schedule(static) num_threads(nthreads)
            for (long i = 0; i < S2; i++) {
                int s1 = (int)(i % S);
                scratch[i] = post_cur[s1] + trans[i];
            }
            /* viterbi.cpp:91-104 */
            #pragma omp parallel for schedule(static) num_threads(nthreads)
            for (int j = 0; j < S; j++) {
                const float *row = scratch + (long)j * S;
                float best = row[0];
                for (int s3 = 1; s3 < S; s3++) {
                    if (row[s3] > best) {
                        best = row[s3];
                        tr_t[j] = s3;
                    }
                }
                post_next[j] = obs_t[j] + best;
            }
        } else {
            #pragma omp parallel for schedule(static) num_threads(nthreads)
            for (int j = 0; j < S; j++) {
                const float *row = trans + (long)j * S;
                float best = post_cur[0] + row[0];
                int32_t arg = 0;
                for (int i = 1; i < S; i++) {
                    float c = post_cur[i] + row[i];
                    if (c > best) { best = c; arg = i; }
                }
                tr_t[j] = arg;
                post_next[j] = obs_t[j] + best;
            }
        }
        float *tmp = post_cur; post_cur = post_next; post_next = tmp;   /* :105-107 */
    }
    memcpy(post_out, post_cur, sizeof(float) * (size_t)S);              /* :111-113 */
}

/*
 * observation  (B,T,S) fp32 row-major      batch_frames (B) int32, 1 <= frames <= T
 * transition   (S,S)  fp32 [next, prev]    initial      (S) fp32
 * indices_out  (B,T)  int32
 * posterior_out (B,S) fp32 or NULL  -- the final posterior rows (viterbi.cpp:204-206)
 */
int torbi_oracle_viterbi_decode(const float *observation, const int32_t *batch_frames,
                                const float *transition, const float *initial,
                                int32_t *indices_out, float *posterior_out,
                                int B, int T, int S, int num_threads, int mode) {
    if (!observation || !batch_frames || !transition || !initial || !indices_out) return ORACLE_EARG;
    if (B < 0 || T < 1 || S < 1) return ORACLE_EARG;
    for (int b = 0; b < B; b++)
        if (batch_frames[b] < 1 || batch_frames[b] > T) return ORACLE_EARG; /* frames==0 is UB upstream (:153) */
    if (num_threads < 1) num_threads = 1;

    float *post_cur = (float *)malloc(sizeof(float) * (size_t)S);
    float *post_next = (float *)malloc(sizeof(float) * (size_t)S);
    float *post_fin = (float *)malloc(sizeof(float) * (size_t)S);
    float *scratch = mode == 0 ? (float *)malloc(sizeof(float) * (size_t)S * S) : NULL;
    int32_t *trellis = (int32_t *)malloc(sizeof(int32_t) * (size_t)T * S);
    if (!post_cur || !post_next || !post_fin || !trellis || (mode == 0 && !scratch)) {
        free(post_cur); free(post_next); free(post_fin); free(scratch); free(trellis);
        return ORACLE_ENOMEM;
    }

    /* batch loop is serial upstream (viterbi.cpp:65); the per-item trellis slab here is
       the b-th (T,S) slice of the reference's (B,T,S) at::zeros tensor (:201-203). */
    for (int b = 0; b < B; b++) {
        const int frames = batch_frames[b];
        const float *obs = observation + (long)b * T * S;
        int32_t *out = indices_out + (long)b * T;
        memset(trellis, 0, sizeof(int32_t) * (size_t)T * S);
        forward_item(obs, transition, initial, post_cur, post_next, scratch, trellis,
                     post_fin, frames, S, mode, num_threads);
        if (posterior_out) memcpy(posterior_out + (long)b * S, post_fin, sizeof(float) * (size_t)S);

        /* posterior.argmax(1) -> first maximal index; repeat over all T columns (:218-221).  ATen's argmax treats
           NaN as the maximum (the FIRST NaN of a row wins); the forward scan above already behaves like the
           reference's on NaN by construction (a NaN candidate at index 0 is never replaced, later ones never win) */
        int32_t arg = 0;
        float best = post_fin[0];
        for (int i = 1; i < S; i++)
            if (post_fin[i] > best || (post_fin[i] != post_fin[i] && best == best)) { best = post_fin[i]; arg = i; }
        for (int t = 0; t < T; t++) out[t] = arg;

        /* viterbi.cpp:153-157 */
        int32_t index = out[frames - 1];
        for (int t = frames - 1; t >= 1; t--) {
            index = trellis[(long)t * S + index];
            out[t - 1] = index;
        }
    }
    free(post_cur); free(post_next); free(post_fin); free(scratch); free(trellis);
    return ORACLE_OK;
}
