/* CPU oracle, plain C restatement of the length-regulator index expansion.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/kokoro_oracle.py header).  Restates
 * `vectorized_expand_tokens` of the reference, src/kokoro/utils/lengths.py:16-96:
 *   dur = clamp(trunc(dur), 0)                    (:31)
 *   len_b = sum_j dur[b][j]; L = max_b len_b clipped to max_len   (:33-36)
 *   L == 0  ->  all-zero output of length max(1,max_len) or 1      (:38-42)
 *   out[b][f] = tok[b][j] with cum[b][j-1] <= f < cum[b][j], zero for f >= len_b (:44-72)
 *   right-pad with zeros to max_len                                 (:74-77)
 * idx[b][f] = j, or -1 for a zero-filled frame.  Integer work: the bar is bit-exact.
 * Pinned by tests/golden/length_regulator.npz (outputs of the reference itself).
 */
#include <stdint.h>

/* Returns L (the output length). idx must hold B*cap entries where cap >= L. */
int64_t kko_length_regulate_index(const int64_t *dur, int64_t B, int64_t P, int64_t max_len /* <0: none */,
                                  int64_t *idx, int64_t idx_stride, int64_t *lens)
{
    int64_t max_expanded = 0;
    for (int64_t b = 0; b < B; ++b) {
        int64_t s = 0;
        for (int64_t j = 0; j < P; ++j) { int64_t d = dur[b * P + j]; s += d > 0 ? d : 0; }
        lens[b] = s;
        if (s > max_expanded) max_expanded = s;
    }
    if (max_len >= 0 && max_expanded > max_len) max_expanded = max_len;
    int64_t L;
    if (max_expanded == 0) L = (max_len >= 0) ? (max_len > 1 ? max_len : 1) : 1;
    else L = (max_len >= 0 && max_len > max_expanded) ? max_len : max_expanded;
    for (int64_t b = 0; b < B; ++b) {
        if (lens[b] > max_expanded) lens[b] = max_expanded;
        int64_t f = 0;
        for (int64_t j = 0; j < P && f < lens[b]; ++j) {
            int64_t d = dur[b * P + j]; if (d < 0) d = 0;
            for (int64_t r = 0; r < d && f < lens[b]; ++r) idx[b * idx_stride + f++] = j;
        }
        for (; f < L; ++f) idx[b * idx_stride + f] = -1;
    }
    return L;
}

/* Gather float payload rows by idx (zero rows where idx < 0). */
void kko_length_regulate_gather(const float *tok, const int64_t *idx, int64_t B, int64_t P, int64_t H,
                                int64_t L, int64_t idx_stride, float *out)
{
    for (int64_t b = 0; b < B; ++b)
        for (int64_t f = 0; f < L; ++f) {
            int64_t j = idx[b * idx_stride + f];
            for (int64_t h = 0; h < H; ++h)
                out[(b * L + f) * H + h] = j < 0 ? 0.0f : tok[(b * P + j) * H + h];
        }
}
