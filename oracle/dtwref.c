/*
 * ORACLE (test infrastructure, never on the product path): CPU restatement of the reference's
 * semi-global dynamic time warping, deepbinner/dtw/dtw.cpp:58-151, for checking the HIP kernel.
 *
 * What it follows:
 *   dtw.cpp:24-26    cost of a cell = (a - b)^2 in fp64
 *   dtw.cpp:29-46    three-way choice: the diagonal wins ties against both others, then the
 *                    strictly smaller of left / up
 *   dtw.cpp:68-87    first column free (cost 0: the query may start anywhere in the reference),
 *                    top row a running sum along the query
 *   dtw.cpp:90-111   the fill
 *   dtw.cpp:113-122  the end: smallest cost in the last column over rows 1 .. ref_len-1, first
 *                    one wins; DBL_MAX and row 0 when ref_len is 1
 *   dtw.cpp:124-145  the walk back to column 0, pairs (ref index, query index) in reverse order
 * One difference, on purpose: an exact left/up tie (diagonal larger than both) is decided by
 * rand() in the reference (dtw.cpp:40-45); here and in the HIP kernel it goes LEFT, so that
 * results are reproducible.  The distance does not depend on that choice, the path may.
 *
 * Pinned against the reference itself: oracle/_ref/dtw.so is dtw.cpp compiled where it lies
 * (oracle/Makefile), tests/test_dtw.py compares the two on seeded inputs.
 */
#include <float.h>
#include <stdint.h>
#include <stdlib.h>

enum { NIL = 0, DIAGONAL = 1, LEFT = 2, UP = 3 };

double dtwref_semi_global(const double* ref, const double* query, int ref_len, int query_len,
                          int* alignment, int* positions, int* path_length) {
    const size_t cells = (size_t)ref_len * (size_t)query_len;
    double* cost = (double*)malloc(cells * sizeof(double));
    unsigned char* path = (unsigned char*)malloc(cells);
#define AT(i, j) ((size_t)(i) * (size_t)query_len + (size_t)(j))
    for (int i = 0; i < ref_len; ++i) {
        cost[AT(i, 0)] = 0.0;
        path[AT(i, 0)] = NIL;
    }
    for (int j = 1; j < query_len; ++j) {
        const double d = query[j] - ref[0];
        cost[AT(0, j)] = cost[AT(0, j - 1)] + d * d;
        path[AT(0, j)] = LEFT;
    }
    for (int i = 1; i < ref_len; ++i) {
        for (int j = 1; j < query_len; ++j) {
            const double diag = cost[AT(i - 1, j - 1)], left = cost[AT(i, j - 1)],
                         top = cost[AT(i - 1, j)];
            double best;
            unsigned char dir;
            if (diag <= left && diag <= top) { dir = DIAGONAL; best = diag; }
            else if (top < left)             { dir = UP;       best = top; }
            else                             { dir = LEFT;     best = left; }
            const double d = ref[i] - query[j];
            cost[AT(i, j)] = best + d * d;
            path[AT(i, j)] = dir;
        }
    }
    double distance = DBL_MAX;
    int end = 0, j = query_len - 1;
    for (int i = 1; i < ref_len; ++i)
        if (cost[AT(i, j)] < distance) { distance = cost[AT(i, j)]; end = i; }
    int i = end, n = 0;
    for (;; ++n) {
        alignment[2 * n] = i;
        alignment[2 * n + 1] = j;
        if (j == 0) break;
        const unsigned char dir = path[AT(i, j)];
        if (dir == DIAGONAL) { --i; --j; }
        else if (dir == LEFT) --j;
        else --i;
    }
#undef AT
    *path_length = n + 1;
    positions[0] = i;
    positions[1] = end;
    free(cost);
    free(path);
    return distance;
}
