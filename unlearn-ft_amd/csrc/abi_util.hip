// Workspace-size queries of the C ABI (include/pdmk.h, SURVEY 8b: "library exposes *_workspace_bytes(dims...) queries and
// allocates nothing persistent").  Every scratch buffer of the hot path is caller-owned; these functions are the single
// source of truth for its size - the Python host (pdm/_pdmk.py) calls them instead of repeating the formulas.
#include "common.h"

extern "C" int64_t pdmk_gemm_splitk_workspace_bytes(int64_t M, int N, int splitk) {
    if (M <= 0 || N <= 0) return -1;
    return (int64_t)(splitk > 1 ? splitk : 1) * M * N * 4;          // [splitk][M][N] fp32 slabs (accumulate = 2)
}

extern "C" int64_t pdmk_groupnorm_workspace_bytes(int B, int G) {   // `ws` of pdmk_groupnorm_fwd / _bwd: B*G*64 doubles
    if (B <= 0 || G <= 0) return -1;
    return (int64_t)B * G * 64 * 8;
}

extern "C" int64_t pdmk_groupnorm_bwd_part_workspace_bytes(int G, int gs) {   // `part_ws`: <= 2048 blocks x 2 x channels floats
    if (G <= 0 || gs <= 0) return -1;
    return (int64_t)2048 * 2 * G * gs * 4;
}

extern "C" int64_t pdmk_layernorm_bwd_part_workspace_bytes(int M, int C) {    // `part_ws`: one (dgamma, dbeta) row per 16 rows
    if (M <= 0 || C <= 0) return -1;
    return ((int64_t)M / 16 + 1) * 2 * C * 4;
}

// `ws` of pdmk_attn_bwd: only cross-attention-like shapes (few keys, many queries) split the query sweep of dK/dV;
// 0 = pass NULL.  16 query splits of [2][B][H][Nk][64] fp32 partials.
extern "C" int64_t pdmk_attn_bwd_workspace_bytes(int B, int H, int Nq, int Nk) {
    if (B <= 0 || H <= 0 || Nq <= 0 || Nk <= 0) return -1;
    if (!(Nk <= 128 && Nq >= 512)) return 0;
    return (int64_t)2 * 16 * B * H * Nk * 64 * 4;
}
