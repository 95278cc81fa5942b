/*
 * deepbinner_hip.h — C ABI of libdeepbinner_hip.so, the MI355X (gfx950) implementation of
 * Deepbinner's classify hot path.
 *
 * The reference has no C ABI on this path: its only device boundary is the Keras call
 *     labels = model.predict(input_signals, batch_size=args.batch_size)   deepbinner/classify.py:361
 * on a model obtained from keras.models.load_model (classify.py:90), wrapped by call_batch
 * (classify.py:325-384).  This header is what a binding for that boundary binds instead; it is
 * modelled on the reference's own ctypes idiom for its one native library
 * (deepbinner/dtw_semi_global.py:30-41: cdll.LoadLibrary, C-contiguous ndpointers, explicit
 * restype/argtypes, caller-allocated outputs, plain-int sizes, scalar return code).
 *
 * Conventions
 *   - every function returns a dbh_status (0 = DBH_OK); nothing throws or exits across the ABI;
 *   - buffers are plain pointers + element counts, C-contiguous, little-endian;
 *   - "_dev" variants take DEVICE pointers and a stream and do not synchronise; the others take
 *     HOST pointers, copy, run and return when the result is in the caller's buffer;
 *   - a model handle belongs to the device that was current when it was created; calls on one
 *     handle must not overlap (the reference host code is single-threaded, classify.py:416-423).
 */
#ifndef DEEPBINNER_HIP_H
#define DEEPBINNER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    DBH_OK = 0,
    DBH_ERR_INVALID_ARGUMENT = 1,
    DBH_ERR_NO_DEVICE = 2,        /* no gfx950-capable HIP device visible */
    DBH_ERR_HIP = 3,              /* a HIP runtime call failed; see dbh_last_error() */
    DBH_ERR_BAD_WEIGHTS = 4,      /* blob size does not match the Deepbinner architecture */
    DBH_ERR_UNSUPPORTED = 5,      /* e.g. input_size != 1024 or n_classes > 32 */
    DBH_ERR_OUT_OF_MEMORY = 6,
    DBH_ERR_COMM = 7              /* RCCL missing or a collective failed; see dbh_comm_last_error() */
} dbh_status;

typedef struct dbh_model dbh_model;      /* opaque: packed weights resident in HBM */
typedef void* dbh_stream;                /* a hipStream_t */
typedef void* dbh_event;                 /* a hipEvent_t */

#define DBH_SIDE_START 0                 /* classify.py:343-344, windows right-padded  (:355) */
#define DBH_SIDE_END 1                   /* classify.py:345-349, windows left-padded   (:357) */
#define DBH_CALL_NONE 0                  /* barcode call 'none' (classify.py:289-290,295)   */
#define DBH_WINDOW 1024                  /* model input size, classify.py:96                */

/* ---- library / device ------------------------------------------------------------------ */
const char* dbh_version(void);
const char* dbh_status_string(int status);
const char* dbh_last_error(void);                       /* text of the last DBH_ERR_HIP      */
int dbh_device_count(int* count);
int dbh_set_device(int ordinal);                        /* replaces set_tensorflow_threads' device_count, classify.py:416-423 */
int dbh_get_device(int* ordinal);
int dbh_device_name(int ordinal, char* buf, int buf_len);
int dbh_device_synchronize(void);

/* ---- raw device memory (so a host language needs no other GPU library) ----------------- */
int dbh_malloc(void** dev_ptr, size_t bytes);
int dbh_free(void* dev_ptr);
int dbh_malloc_host(void** host_ptr, size_t bytes);     /* pinned, for overlapped H2D        */
int dbh_free_host(void* host_ptr);
/* The address the GPU reaches a pinned host buffer at.  The *_dev entry points take it wherever
 * they take a device pointer for RESULTS (calls, probabilities): the kernels then store them
 * straight into host memory - no copy command between two launches of a stream (one GPU's calls
 * in bench.py; 40 KB per 10,000 reads). */
int dbh_host_device_pointer(void* host_ptr, void** dev_ptr);
int dbh_memcpy_h2d(void* dev_dst, const void* host_src, size_t bytes, dbh_stream stream);
int dbh_memcpy_d2h(void* host_dst, const void* dev_src, size_t bytes, dbh_stream stream);
int dbh_memcpy_d2d(void* dev_dst, const void* dev_src, size_t bytes, dbh_stream stream);
int dbh_stream_create(dbh_stream* stream);
int dbh_stream_destroy(dbh_stream stream);
int dbh_stream_synchronize(dbh_stream stream);
int dbh_event_create(dbh_event* event);
int dbh_event_destroy(dbh_event event);
int dbh_event_record(dbh_event event, dbh_stream stream);
int dbh_event_synchronize(dbh_event event);
/* what is queued on `stream` after this call waits until `event` (recorded on another stream of
 * the device) has happened: the edge between a classification stream and the side stream that
 * gathers and copies its calls (bench.py, sharding.SideGather) */
int dbh_stream_wait_event(dbh_stream stream, dbh_event event);
int dbh_event_elapsed_ms(dbh_event start, dbh_event stop, float* ms);

/* ---- model (replaces keras.models.load_model, classify.py:90) -------------------------- */
/* weights: the canonical flat fp32 blob (deepbinner_amd/model_format.py): for conv 1..20
 * kernel[k][C_in][C_out] then bias[C_out]; for batch-norm 1..7 gamma, beta, mean, variance.
 * n_floats must equal the architecture's parameter count for n_classes (107,197 for 13). */
int dbh_model_create(const float* weights, int64_t n_floats, int n_classes, int input_size,
                     dbh_model** model);
int dbh_model_destroy(dbh_model* model);
int dbh_model_input_size(const dbh_model* model, int* input_size);   /* model.inputs[0].shape[1], classify.py:93-96  */
int dbh_model_output_size(const dbh_model* model, int* n_classes);   /* model.outputs[0].shape[1], classify.py:94-97 */

/* The host-buffer entry points (dbh_predict, dbh_classify_i16) make the model's device the calling
 * thread's current device (HIP keeps that per thread), so a model may be driven from any thread -
 * two models from two threads at once, each on its own streams and staging buffers.  One model is
 * for one caller at a time.  The *_dev entry points work on the caller's device pointers and
 * leave the current device alone; launches of one model queued on DIFFERENT streams may overlap
 * on the device (the per-launch scratch is kept per stream), the host-side calls themselves must
 * still not overlap. */
/* ---- seam b1: model.predict (classify.py:361) ------------------------------------------ */
/* x: n_windows x 1024 fp32 (already normalised);  probs: n_windows x n_classes fp32 softmax. */
int dbh_predict(dbh_model* model, const float* x_host, int64_t n_windows, float* probs_host);
int dbh_predict_dev(dbh_model* model, const float* x_dev, int64_t n_windows, float* probs_dev,
                    dbh_stream stream);

/* ---- seam b2: call_batch (classify.py:325-384) ------------------------------------------ */
/* samples: concatenated int16 raw signals; offsets[n_reads+1] delimit each read (any length
 * >= 0).  For each read: scan_size/512 windows (classify.py:330-349), normalise
 * (trim_signal.py:61-69, fp64), zero-pad (classify.py:352-357), forward pass, min/max merge
 * (classify.py:368-374), make_sum_to_one (classify.py:387-393) and the top-2 threshold call
 * (classify.py:285-295).  probs: n_reads x n_classes fp32; calls: n_reads int32, 0 = 'none'. */
int dbh_classify_i16(dbh_model* model, const int16_t* samples_host, const int64_t* offsets_host,
                     int64_t n_reads, int side, int scan_size, double score_diff,
                     float* probs_host, int32_t* calls_host);
/* Host buffers that are pinned (dbh_malloc_host / dbh_host_alloc) are read by the GPU's DMA engine
 * where they lie; pageable ones go through a pinned staging copy first.  Either way the reads
 * travel in groups (32,768 windows per model by default: dbh_model_set_host_group) through three
 * slots, so that the upload of one group, the kernels of another and the results of a third
 * overlap. */
int dbh_model_set_host_group(dbh_model* model, int64_t windows_per_group /* 0 = default */);
/* The forward kernel is persistent: one workgroup per CU walks the windows of a launch.  A model
 * whose launches share the GPU with the inflate kernels of the next containers (the streaming
 * path: dbh_classify_pair_deflated on several queues) leaves them n_cus CUs - otherwise those
 * kernels find no CU free until the launch ends, and the workgroups they displaced start late.
 * 0 = all CUs again. */
int dbh_model_reserve_cus(dbh_model* model, int n_cus);

/* The body of the reference's per-batch loop (classify.py:141-171) for one batch of reads and BOTH
 * models in one call: the samples are uploaded once, the start model scans the first and the end
 * model the last scan_size samples of every read (call_batch with side 'start' / 'end',
 * classify.py:325-384), and combine_calls (classify.py:298-322, DBH_REQUIRE_*) gives the final
 * call - all queued on one stream per group, nothing but the calls (and whatever else is asked
 * for) coming back.  Either model may be NULL: calls_host is then the other model's calls.
 * calls_host: n_reads final calls.  Optional (NULL = not wanted): the per-side calls (n_reads
 * int32 each) and per-side probabilities (n_reads x n_classes fp32 each) - what --verbose
 * prints.  Both models must live on the same device and agree on n_classes. */
int dbh_classify_pair_i16(dbh_model* start_model, dbh_model* end_model,
                          const int16_t* samples_host, const int64_t* offsets_host,
                          int64_t n_reads, int scan_size, double score_diff, int combine_mode,
                          int32_t* calls_host, int32_t* start_calls_host, int32_t* end_calls_host,
                          float* start_probs_host, float* end_probs_host);

/* Pinned host memory behind the allocator signature libdeepbinner_fast5.so takes
 * (deepbinner_fast5.h: f5_set_sample_allocator): the loader's threads then write a batch straight
 * into memory the entry points above upload without a staging copy.  `user` is ignored. */
void* dbh_host_alloc(size_t bytes, void* user);
void dbh_host_release(void* ptr, void* user);
int dbh_host_is_pinned(const void* ptr, size_t bytes, int* pinned);

/* Optional hint for the *_dev entry points: "every read in the sample buffer is read_length
 * samples long and the buffer holds at least capacity_samples samples".  The forward kernel then
 * requests a read's samples from read_index * read_length TOGETHER with offsets[read_index]
 * instead of after it (a dependent load costs ~3k cycles at the top of a kernel) and fetches
 * again from the true place wherever the offsets disagree, so a wrong hint costs time, never
 * correctness; speculative addresses beyond capacity_samples are not touched.  read_length 0
 * clears the hint.  dbh_classify_i16 (host buffers) works this out per group by itself. */
int dbh_model_set_read_length_hint(dbh_model* model, int64_t read_length,
                                   int64_t capacity_samples);
/* device-resident variant; workspace_dev must hold dbh_classify_workspace_bytes() bytes (unused,
 * and may be NULL, when scan_size is 512: the whole read then finishes inside one launch). */
int dbh_classify_workspace_bytes(const dbh_model* model, int64_t n_reads, int scan_size,
                                 size_t* bytes);
int dbh_classify_i16_dev(dbh_model* model, const int16_t* samples_dev, const int64_t* offsets_dev,
                         int64_t n_reads, int side, int scan_size, double score_diff,
                         float* probs_dev, int32_t* calls_dev, void* workspace_dev,
                         dbh_stream stream);

/* The same job for many reads in batches of batch_size reads (the reference's --batch_size loop,
 * classify.py:130), queued in order on `stream`.  offsets_dev holds n_reads+1 ABSOLUTE offsets
 * into samples_dev.  With scan_size 512 every batch is a single kernel launch (slice + normalise
 * + CNN + renormalise + call); otherwise two (CNN, merge).  Does not block the host. */
int dbh_classify_i16_batched_dev(dbh_model* model, const int16_t* samples_dev,
                                 const int64_t* offsets_dev, int64_t n_reads, int batch_size,
                                 int side, int scan_size, double score_diff, float* probs_dev,
                                 int32_t* calls_dev, dbh_stream stream);

/* combine_calls (classify.py:298-322) for whole arrays of start-model and end-model calls on the
 * device: out[i] is the final call of read i (DBH_CALL_NONE or a barcode number).  Agreement
 * stands; otherwise REQUIRE_BOTH refuses, REQUIRE_START keeps a start call the end model is silent
 * on, REQUIRE_EITHER (the default, deepbinner.py:315-316) keeps whichever side called when the
 * other is silent.  out_dev may alias either input.  Does not block the host. */
#define DBH_REQUIRE_EITHER 0
#define DBH_REQUIRE_START 1
#define DBH_REQUIRE_BOTH 2
int dbh_combine_calls_dev(const int32_t* start_calls_dev, const int32_t* end_calls_dev,
                          int64_t n_reads, int mode, int32_t* out_dev, dbh_stream stream);

/* ---- compressed input: the Signal chunks of fast5 files, inflated on the GPU ---------------- */
/* The reference reads `Signal[:]` through h5py (load_fast5s.py:33-43): libhdf5 runs zlib's
 * inflate() over every chunk on the host.  Here the loader may hand over the chunks AS STORED
 * (zlib streams, RFC 1950/1951: the HDF5 "deflate" filter) and the GPU inflates them - thousands
 * of streams side by side, bit-exact with zlib, same accept / reject decisions (tests/
 * test_inflate.py).  A stream is described by where its bytes lie in the compressed buffer and
 * where its output goes in the output buffer; out_bytes is how much of its output is wanted: a
 * stream that holds more is cut there (a partial last chunk), one that ends earlier is
 * zero-extended (MinKNOW's short final chunk, as libhdf5 does it).  DBH_INFLATE_STORED: the
 * bytes are the data itself (an unfiltered chunk, or what the host inflated for the filters the
 * GPU does not do) - copied, zero-extended.
 * status per stream: 0 = ok; anything else = damaged or beyond this decoder (its output is then
 * all zeros): the caller decodes that stream on the host if it wants the verdict of zlib itself.
 * The compressed buffer must be readable for 64 bytes beyond its end (the decoder fetches ahead).
 * Output regions must not overlap; out_offset must be even (samples are int16). */
#define DBH_INFLATE_ZLIB 0
#define DBH_INFLATE_STORED 1
typedef struct dbh_inflate_stream {
    int64_t comp_offset, comp_bytes;       /* the stream inside the compressed buffer           */
    int64_t out_offset, out_bytes;         /* its output inside the output buffer (bytes)       */
    int32_t mode, reserved;
} dbh_inflate_stream;
const char* dbh_inflate_last_error(void);
/* total_out_bytes = size of the output buffer the streams write into */
int dbh_inflate_workspace_bytes(int64_t total_out_bytes, int64_t n_streams, size_t* bytes);
/* comp_bytes = size of the compressed buffer (the decoder's read-ahead stops 64 bytes behind it).
 * Kernel 1 (Huffman codes -> tokens) gives every stream a wavefront of its own, in the order of
 * the records (the longest first, if the caller can: the launch then ends evenly).
 * streams_per_lane only matters to the kernel's older form (environment DEEPBINNER_INFLATE_KERNEL=
 * lane: one LANE per stream; 0 = 1): its lanes take streams off a counter - with one stream per
 * lane a launch is as wide as the streams are many and lasts as long as the longest of them; with
 * n, a launch 1/n as wide does the same work, and lasts no longer if the long streams come first
 * in the records and are long enough.
 * Kernel 2 (tokens -> bytes) reads the output buffer it writes (a match whose source lies more
 * than 8 KiB back is copied from the stream's own flushed output): out_dev must not be mapped
 * write-combined or read-protected.  DEEPBINNER_INFLATE_RESOLVE=rounds selects its older form
 * (the whole 32 KiB window in LDS).  The two kernels run as ONE launch, a pair of waves per
 * stream, kernel 2 resolving a stream's tokens while kernel 1 still decodes it
 * (DEEPBINNER_INFLATE_PAIR=0: two launches). */
int dbh_inflate_dev(const uint8_t* comp_dev, int64_t comp_bytes,
                    const dbh_inflate_stream* streams_dev, int64_t n_streams,
                    int64_t total_out_bytes, uint8_t* out_dev, void* workspace_dev,
                    int32_t* status_dev, int streams_per_lane, dbh_stream stream);
/* host buffers in, host buffers out (tests, tools); kernel_ms (may be NULL): the two kernels */
int dbh_inflate(const uint8_t* comp_host, size_t comp_bytes, const dbh_inflate_stream* streams_host,
                int64_t n_streams, uint8_t* out_host, size_t out_bytes, int32_t* status_host,
                int streams_per_lane, double* kernel_ms);

/* The whole of a batch from stored chunks to barcode calls in one call: upload of the compressed
 * bytes (in place if they are pinned: the native loader's raw batches are), inflate, the start
 * model over the first and the end model over the last scan_size samples of every read,
 * combine_calls - dbh_classify_pair_i16 for reads that are still deflated.  offsets_host (n_reads
 * + 1, in samples, starting at 0): where each read's signal lies once decoded; the streams'
 * out_offset / out_bytes address the same buffer in bytes.  comp_host must be readable for 64
 * bytes beyond comp_bytes (a pageable buffer is copied and padded, so it need not be).
 * Calls from several threads (one model pair each: the queues of the streaming path) share the
 * device: the inflating of one runs beside the classification of another, but the forward
 * launches of all of them go through ONE stream per device, one launch at a time - the forward
 * kernel is persistent, and two of them on one GPU only stretch each other and everything queued
 * behind them (DEEPBINNER_FORWARD_STREAM=own: each call on its own stream, for comparison).
 * stream_status_host (n_streams, may be NULL): the decoder's verdict per stream - a read one of
 * whose streams failed was classified on zeros; the caller decodes it on the host and asks
 * again.  samples_host (may be NULL): the decoded signals, all of them (realtime's binning).
 * stage_ms (may be NULL, else 3 doubles): upload, inflate, classify in milliseconds on the
 * device (HIP events; tools). */
int dbh_classify_pair_deflated(dbh_model* start_model, dbh_model* end_model,
                               const uint8_t* comp_host, int64_t comp_bytes,
                               const dbh_inflate_stream* streams_host, int64_t n_streams,
                               const int64_t* offsets_host, int64_t n_reads, int scan_size,
                               double score_diff, int combine_mode, int32_t* calls_host,
                               int32_t* stream_status_host, int16_t* samples_host,
                               double* stage_ms);
/* The same call for the reference's verbose table (classify.py:157-171 prints, beside the final
 * call, each side's 2-decimal probabilities and - with two models - each side's own call):
 * optional outputs (NULL = not wanted) as in dbh_classify_pair_i16 - the per-side calls (n_reads
 * int32 each) and per-side merged probabilities (n_reads x n_classes fp32 each).  A side whose
 * model is NULL leaves its arrays untouched. */
int dbh_classify_pair_deflated_verbose(dbh_model* start_model, dbh_model* end_model,
                                       const uint8_t* comp_host, int64_t comp_bytes,
                                       const dbh_inflate_stream* streams_host, int64_t n_streams,
                                       const int64_t* offsets_host, int64_t n_reads, int scan_size,
                                       double score_diff, int combine_mode, int32_t* calls_host,
                                       int32_t* stream_status_host, int16_t* samples_host,
                                       double* stage_ms, int32_t* start_calls_host,
                                       int32_t* end_calls_host, float* start_probs_host,
                                       float* end_probs_host);

/* ---- multi-device: reads shard over the GPUs of one node, calls are all-gathered ---------- */
/* The reference is single-device (its only knob: set_tensorflow_threads, classify.py:416-423).
 * Reads are independent, so every GPU classifies a contiguous shard with its own model replica
 * (dbh_model_create under dbh_set_device) and no data-path collective; the one exchange is an
 * all-gather of the per-read int32 calls over RCCL / xGMI.  Two host
 * models, same entry points:
 *   one process, n devices:  dbh_comm_init_all (ncclCommInitAll); the arrays handed to
 *                            dbh_comm_all_gather_i32 have one entry per device;
 *   one process per GPU:     rank 0 calls dbh_comm_unique_id and ships the DBH_COMM_ID_BYTES bytes
 *                            to the other ranks over any host channel; every rank then calls
 *                            dbh_comm_init_rank with ITS device current; arrays have one entry.
 * librccl.so.1 is looked up at the first of these calls (dlopen), not at load time. */
typedef struct dbh_comm dbh_comm;
#define DBH_COMM_ID_BYTES 128
#define DBH_COMM_RCCL 0                  /* ncclAllGather                                        */
#define DBH_COMM_COPY 1                  /* hipMemcpy(Peer)Async copies, single-process form only:
                                            for boxes where two "devices" are one physical GPU,
                                            which RCCL refuses                                    */
int dbh_comm_available(void);            /* 1 if librccl could be loaded                          */
const char* dbh_comm_last_error(void);
int dbh_comm_init_all(int n_devices, const int* ordinals /* NULL = 0..n-1 */, int transport,
                      dbh_comm** comm);
int dbh_comm_unique_id(void* id_out /* DBH_COMM_ID_BYTES */);
int dbh_comm_init_rank(const void* id, int n_ranks, int rank, dbh_comm** comm);
int dbh_comm_info(const dbh_comm* comm, int* n_ranks, int* n_local, int* transport);
/* recv_dev[i] (n_ranks * count int32 on local device i) = the send_dev blocks of all ranks in rank
 * order; queued on streams[i] behind whatever produced send_dev[i]; does not block the host. */
int dbh_comm_all_gather_i32(dbh_comm* comm, const int32_t* const* send_dev,
                            int32_t* const* recv_dev, int64_t count, const dbh_stream* streams);
int dbh_comm_destroy(dbh_comm* comm);

/* ---- pieces of seam b2, exposed for parity tests ---------------------------------------- */
/* windows_dev: (n_reads * steps) x 1024 fp32, read-major (window index = read*steps + step). */
int dbh_normalise_windows_dev(const int16_t* samples_dev, const int64_t* offsets_dev,
                              int64_t n_reads, int side, int scan_size, float* windows_dev,
                              dbh_stream stream);
int dbh_merge_calls_dev(const float* window_probs_dev, int64_t n_reads, int steps, int n_classes,
                        double score_diff, float* probs_dev, int32_t* calls_dev,
                        dbh_stream stream);

/* ---- introspection ------------------------------------------------------------------------ */
/* Activations after stage 'A'..'G'  for n_windows windows, row-major
 * [window][position][channel]; out_host must hold n_windows * dbh_stage_floats(stage) floats. */
int dbh_stage_floats(int stage, int64_t* floats_per_window);
int dbh_debug_forward(dbh_model* model, const float* x_host, int64_t n_windows, int stage,
                      float* out_host);
/* name and static resource use of the forward kernel, for bench/roofline bookkeeping */
int dbh_forward_kernel_info(int* threads_per_block, int* lds_bytes, int* vgprs);
/* Matrix instructions (v_mfma_f32_16x16x4_f32, 2,048 FLOP each) the forward kernel ISSUES per
 * window for a model of n_classes classes - fewer than the 33,629,952 algorithmic FLOP of the
 * direct convolutions because six layers run as Winograd F(4,3) / F(2,3).  A constant of the build
 * (dbh_layout.h), checked against rocprofv3's SQ_INSTS_MFMA in profiles/. */
int dbh_forward_executed_mfmas(int n_classes, int64_t* mfmas_per_window,
                               int64_t* flop_per_window);
/* Run the forward kernel only up to and including stage last_stage (0 = 'A' .. 6 = 'G'), writing
 * nothing: lets a profiler attribute kernel time to stages by differencing. */
int dbh_forward_truncated_dev(dbh_model* model, const float* x_dev, int64_t n_windows,
                              int last_stage, dbh_stream stream);
/* Timeline of one forward launch: every wave of every workgroup stamps the shader cycle counter at
 * each phase boundary (slot meanings: tools/timeline.py).  stamps_host: n_windows x 8 x 64 int64. */
int dbh_forward_timeline(dbh_model* model, const float* x_host, int64_t n_windows,
                         int64_t* stamps_host);
/* The same for the fused seam-b2 mode: n_reads reads of exactly 1,024 int16 samples each (side
 * start, one scan step), so that the slice + normalise front of stage A is on the timeline. */
int dbh_forward_timeline_i16(dbh_model* model, const int16_t* samples_host, int64_t n_reads,
                             int64_t* stamps_host);
/* Live kernel timing: with enable = n > 0, every n-th launch of the forward kernel is bracketed by
 * HIP events on the stream it is launched on (n = 1: every launch; the event pair itself costs a
 * few microseconds of queue time, so sampling keeps the measurement from slowing what it
 * measures); dbh_forward_timing_read synchronises those events, returns the summed kernel time,
 * the number of timed launches and of windows they covered, and resets the tally. 0 = off. */
int dbh_forward_timing_enable(dbh_model* model, int enable);
/* The same with ONE event pair around a run of `span` consecutive launches (span <= every_nth)
 * starting at every every_nth-th launch: the pair's own cost (~2.5 us of dispatch latency lands
 * inside the bracket) is shared by the run, so total_ms / launches is the launch-to-launch period
 * of back-to-back launches, which is what rocprofv3's per-kernel duration plus the dispatch gap
 * adds up to.  Brackets still open at dbh_forward_timing_read are dropped. */
int dbh_forward_timing_enable_span(dbh_model* model, int every_nth, int span);
int dbh_forward_timing_read(dbh_model* model, double* total_ms, int64_t* launches,
                            int64_t* windows);

/* Clock probe.  The MFMA peak of the data sheet assumes 2.4 GHz; under this kernel's load the
 * shader clock runs lower (power management), and "how busy is the matrix pipe" is a ratio of
 * CYCLES.  With the probe on, every workgroup of a production launch of the forward kernel notes
 * the shader clock counter (s_memtime) and the constant-rate wall clock (s_memrealtime,
 * hipDeviceAttributeWallClockRate) at its start and at its end (two stores per workgroup);
 * dbh_forward_clock_read synchronises the device and returns the median ratio of the model's
 * latest launch as a frequency in GHz. */
int dbh_forward_clock_enable(dbh_model* model, int enable);
int dbh_forward_clock_read(dbh_model* model, double* shader_ghz);
/* Phase stamps, asked for on top of the clock probe (dbh_forward_phases_enable; each stamp is a
 * counter read and an LDS word by one lane - ~25 per group, about 1 % of the kernel's time, so they are
 * not part of a timed run): the forward kernel keeps, in LDS, the shader clock at five points of each
 * of a workgroup's first 12 groups of windows (it takes its windows four at a time): the group's
 * start, the end of its stage A-C loop (conv1d_1 .. conv1d_7 of each window), of the stage D-E chain
 * (conv1d_8 .. conv1d_16 of the four together), of its stage F loop (conv1d_17 of each window) and of
 * the batched tail - one lane, right behind a barrier: the kernel measured is the kernel that ships.
 * dbh_forward_phases_read returns the mean cycles of the five intervals (the fifth runs to the next
 * group's start; tail-less groups leave it near zero) over the steady-state groups of the model's
 * latest launch, and how many groups that was.
 * mean_cycles_14[5..8]: stage F's inner intervals as its first wave sees them, summed over the group's
 * windows: to the end of conv1d_17's MFMAs, to behind its barrier, to the end of the reduction, to
 * behind the group's last barrier.  [9..11]: stages A-C's, summed over the group's windows: a window's
 * top to the start of stage B's chain, the chain, conv1d_7 (of all windows but the group's last);
 * [12], [13]: the group's last conv1d_7, and from its end to the start of the stage D-E chain. */
int dbh_forward_phases_enable(dbh_model* model, int enable);
int dbh_forward_phases_read(dbh_model* model, double* mean_cycles_14, int64_t* groups);

#ifdef __cplusplus
}
#endif
#endif /* DEEPBINNER_HIP_H */
