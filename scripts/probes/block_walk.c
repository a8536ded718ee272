/* A FAMILY of fixed-block walks over the affine-gap global alignment matrix, for measuring how much alignment::realign's call can depend on block-aligner's heuristic
 * (alignment.rs:14,44-52: Block::align with size 8..=8, no x-drop, no traceback; crate block-aligner, not vendored and not available offline: its direction / tie /
 * step rules cannot be restated with certainty, so every plausible combination is run: scripts/block_walk_family.py).
 * Scores: match +1, mismatch -1 (NW1), a gap of length k costs open + (k - 1) * extend with open = -2, extend = -1 (Gaps { open: -2, extend: -1 }).
 * The walk: a B x B block starts at the top-left corner; it is shifted right or down by `step` cells at a time (the new B x step strip is computed from its computed
 * neighbours; cells never computed count as -infinity), direction chosen by comparing the block's right border column with its bottom border row
 * (rule 0: maxima, rule 1: sums) with ties going right (tie 0) or down (tie 1); at a matrix edge the only possible direction is taken; the score is D[n][m].
 * gcc -O2 -shared -fPIC -o block_walk.so block_walk.c */
#include <stdint.h>
#include <string.h>
#define NEG (-1000000)
static int mx(int a, int b) { return a > b ? a : b; }
typedef struct { int D[65][65], C[65][65], R[65][65]; uint8_t done[65][65]; } Mat;   /* D best, C gap-in-column state (vertical), R gap-in-row state (horizontal) */
static void cell(Mat* M, const uint8_t* q, const uint8_t* r, int i, int j, int open, int ext) {
    if (M->done[i][j]) return;
    int d = NEG, c = NEG, h = NEG;
    if (i == 0 && j == 0) { d = 0; }
    else {
        if (i > 0 && M->done[i - 1][j]) c = mx(M->D[i - 1][j] + open, M->C[i - 1][j] + ext);
        if (j > 0 && M->done[i][j - 1]) h = mx(M->D[i][j - 1] + open, M->R[i][j - 1] + ext);
        if (i > 0 && j > 0 && M->done[i - 1][j - 1]) d = M->D[i - 1][j - 1] + (q[i - 1] == r[j - 1] ? 1 : -1);
        d = mx(d, mx(c, h));
    }
    if (d < NEG / 2) d = NEG;
    if (c < NEG / 2) c = NEG;
    if (h < NEG / 2) h = NEG;
    M->D[i][j] = d; M->C[i][j] = c; M->R[i][j] = h; M->done[i][j] = 1;
}
int exact_score(const uint8_t* q, const uint8_t* r, int n, int m, int open, int ext) {
    static __thread Mat M; memset(M.done, 0, sizeof M.done);
    for (int i = 0; i <= n; ++i) for (int j = 0; j <= m; ++j) cell(&M, q, r, i, j, open, ext);
    return M.D[n][m];
}
/* rows / columns are 1-based cells of the matrix (row 0 / column 0 = the gap borders, computed with the first block) */
int walk_score(const uint8_t* q, const uint8_t* r, int n, int m, int B, int step, int rule, int tie, int open, int ext, int* cells_out) {
    static __thread Mat M; memset(M.done, 0, sizeof M.done);
    int i0 = 0, j0 = 0;                                  /* the block covers rows i0 .. i0 + B, columns j0 .. j0 + B (clipped) */
    for (int i = 0; i <= (B < n ? B : n); ++i) for (int j = 0; j <= (B < m ? B : m); ++j) cell(&M, q, r, i, j, open, ext);
    for (;;) {
        const int ie = i0 + B < n ? i0 + B : n, je = j0 + B < m ? j0 + B : m;
        if (ie == n && je == m) break;
        int dir;                                          /* 0 right, 1 down */
        if (je == m) dir = 1; else if (ie == n) dir = 0;
        else {
            long a = rule ? 0 : NEG, b = rule ? 0 : NEG;
            for (int i = i0; i <= ie; ++i) { const int v = M.done[i][je] ? M.D[i][je] : NEG; if (rule) a += v; else a = v > a ? v : a; }     /* right border */
            for (int j = j0; j <= je; ++j) { const int v = M.done[ie][j] ? M.D[ie][j] : NEG; if (rule) b += v; else b = v > b ? v : b; }     /* bottom border */
            dir = b > a ? 1 : (a > b ? 0 : tie);
        }
        if (dir == 0) { j0 += step; if (j0 + B > m) j0 = m - B > 0 ? m - B : 0; }
        else          { i0 += step; if (i0 + B > n) i0 = n - B > 0 ? n - B : 0; }
        const int ie2 = i0 + B < n ? i0 + B : n, je2 = j0 + B < m ? j0 + B : m;
        for (int i = i0; i <= ie2; ++i) for (int j = j0; j <= je2; ++j) cell(&M, q, r, i, j, open, ext);
    }
    if (cells_out) { int c = 0; for (int i = 0; i <= n; ++i) for (int j = 0; j <= m; ++j) c += M.done[i][j]; *cells_out = c; }
    return M.D[n][m];
}
void batch(const uint8_t* Q, const uint8_t* R, int N, int n, int m, int B, int step, int rule, int tie, int* out) {      /* B == 0: the exact DP */
    for (int x = 0; x < N; ++x) out[x] = B ? walk_score(Q + (long)x * n, R + (long)x * m, n, m, B, step, rule, tie, -2, -1, 0) : exact_score(Q + (long)x * n, R + (long)x * m, n, m, -2, -1);
}
