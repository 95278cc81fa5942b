/*
 * deepbinner_dtw.h - C ABI of libdeepbinner_dtw.so: semi-global dynamic time warping on an
 * MI355X (gfx950), SURVEY.md §8f rank 4.
 *
 * Replaces the reference's one native function,
 *   deepbinner/dtw/dtw.h:17-18     double semi_global_dtw(ref, query, ref_len, query_len,
 *                                                        alignment, positions, path_length)
 *   deepbinner/dtw/dtw.cpp:58-151  its implementation (fp64 cost matrix, direction matrix, walk back)
 * which deepbinner/dtw_semi_global.py:30-41 binds with ctypes (LoadLibrary, ndpointer arguments,
 * restype double).  `semi_global_dtw` below has the same name, arguments and results, so that
 * binding works on this library unchanged; `dtw_semi_global_batch` is the form that fills a GPU:
 * many (reference signal, query signal) pairs per call, one wavefront each.
 *
 * Arithmetic: fp64 subtract, multiply, add, compare in the order of dtw.cpp (no fused
 * multiply-add), so distances are bit-identical to the reference's.  One documented difference: an
 * exact tie between LEFT and UP (with the diagonal larger than both) is broken by rand() in the
 * reference (dtw.cpp:40-45) and goes LEFT here; the distance does not depend on it.
 *
 * All pointers are host pointers; the calls block.  Thread safe (one call at a time runs on the
 * device).  Errors: `semi_global_dtw` returns NaN and leaves path_length[0] = 0;
 * `dtw_semi_global_batch` returns a status.
 */
#ifndef DEEPBINNER_DTW_H
#define DEEPBINNER_DTW_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DTW_OK 0
#define DTW_ERR_ARGUMENT 1     /* null pointer, non-positive length, offsets not ascending */
#define DTW_ERR_NO_DEVICE 2    /* no HIP device / not gfx950 */
#define DTW_ERR_HIP 3          /* a HIP call failed: dtw_last_error() has the text */

const char* dtw_version(void);
const char* dtw_status_string(int status);
const char* dtw_last_error(void);

/* dtw.h:17-18.  alignment: room for 2 * (ref_len + query_len) ints; receives path_length[0] pairs
 * (ref index, query index) from the END of the alignment backwards.  positions[0] / positions[1]:
 * first and last reference index of the aligned region.  Returns the distance (the smallest
 * accumulated cost in the last query column over reference rows 1 .. ref_len-1). */
double semi_global_dtw(const double* ref, const double* query, int ref_len, int query_len,
                       int* alignment, int* positions, int* path_length);

/* n_pairs alignments in one call.  Pair p aligns refs[ref_offsets[p] .. ref_offsets[p+1]) with
 * queries[query_offsets[p] .. query_offsets[p+1]) (both offset arrays have n_pairs + 1 entries,
 * every pair needs at least one sample of each).  Outputs: distances[p]; positions[2p], [2p+1];
 * path_lengths[p]; and, unless `alignment` is NULL, the pairs of p from
 * alignment[2 * (ref_offsets[p] + query_offsets[p])] on, laid out as for semi_global_dtw. */
int dtw_semi_global_batch(const double* refs, const int64_t* ref_offsets, const double* queries,
                          const int64_t* query_offsets, int64_t n_pairs, double* distances,
                          int32_t* positions, int32_t* path_lengths, int32_t* alignment);

/* Device time of the kernels of the last dtw_semi_global_batch / semi_global_dtw call on this
 * thread's device, in milliseconds (HIP events around the launches), and the number of matrix
 * cells they filled. */
int dtw_last_kernel_time(double* milliseconds, int64_t* cells);

#ifdef __cplusplus
}
#endif
#endif
