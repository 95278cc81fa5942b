/*
 * deepbinner_fast5.h - C ABI of libdeepbinner_fast5.so, the native (host-only, C++) fast5 loader
 * in front of the classify path.
 *
 * Replaces, for the one thing the classify path needs from a fast5 file - the read id and the raw
 * int16 signal - the reference's h5py calls:
 *   deepbinner/load_fast5s.py:25-49   get_read_id_and_signal   (f5_open + f5_read_info +
 *                                                               f5_read_signal, or f5_load_batch)
 *   deepbinner/load_fast5s.py:93-102  get_root_level_keys      (f5_layout)
 *   deepbinner/classify.py:141-150    the per-batch loading loop (f5_load_batch: worker threads,
 *                                                               packed scan regions out)
 * It implements the same slice of the HDF5 file format as deepbinner_amd/hdf5_lite.py (superblock
 * v0-v3, object headers v1/v2, symbol-table / compact / dense groups, attributes incl. dense
 * storage and variable-length strings, compact / contiguous / chunked datasets with deflate,
 * shuffle and fletcher32 filters, chunk indexes of layout versions 1-4: v1 B-tree, single chunk,
 * implicit, fixed array, extensible array) and nothing else; every offset taken from the file is
 * bounds checked.  No HIP, no Python: plain pointers and sizes.
 *
 * Threading: f5_file handles are not shared between threads; f5_load_batch runs its own
 * worker threads and is itself safe to call from several threads at once.
 */
#ifndef DEEPBINNER_FAST5_H
#define DEEPBINNER_FAST5_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define F5_OK 0
#define F5_ERR_OPEN 1      /* cannot open or map the file */
#define F5_ERR_FORMAT 2    /* not HDF5, damaged, or outside the supported subset */
#define F5_ERR_NO_READ 3   /* no read in the file / index out of range / no read_id or Signal */
#define F5_ERR_MULTI 4     /* several reads where the caller asked for a one-read file */
#define F5_ERR_ARGUMENT 5
#define F5_ERR_FILTER 6    /* Signal compressed with a filter other than deflate / shuffle /
                              fletcher32 (e.g. ONT's VBZ, HDF5 filter 32020) */
#define F5_ERR_EXISTS 7    /* f5_write_single_reads: a file of that name is there already - left
                              as it is (the reference never moves a file over another one,
                              realtime.py:111-144: such a clash is counted and skipped) */

#define F5_READ_ID_MAX 64  /* bytes per read id slot, NUL terminated (a read id is a 36-char UUID) */

/* Layouts, as the reference tells them apart (load_fast5s.py:29-43, 67-90). */
#define F5_LAYOUT_NONE 0         /* neither /Raw nor read_* at the root */
#define F5_LAYOUT_SINGLE_OLD 1   /* /Raw/Reads/<one group>           */
#define F5_LAYOUT_SINGLE_NEW 2   /* exactly one /read_<id>/Raw        */
#define F5_LAYOUT_MULTI 3        /* several /read_<id>/Raw            */

typedef struct f5_file f5_file;

const char* f5_version(void);
const char* f5_status_string(int status);
/* Hardware threads this process can keep busy: online CPUs, affinity mask and cgroup CPU quota
 * taken together - what "one thread per hardware thread" (n_threads <= 0) means below. */
int f5_usable_cpus(void);

int f5_open(const char* path, f5_file** out);
void f5_close(f5_file* file);
/* Layout and number of reads (1 for the single-read layouts). */
int f5_layout(f5_file* file, int* layout, int64_t* n_reads);
/* Read `index` (reads are ordered by group name, as h5py lists them): its id and signal length. */
int f5_read_info(f5_file* file, int64_t index, char read_id[F5_READ_ID_MAX], int64_t* n_samples);
/* Samples [first, first + count) of read `index`; only the chunks that overlap are inflated. */
int f5_read_signal(f5_file* file, int64_t index, int64_t first, int64_t count, int16_t* out);

/* One-read files -> packed signals, loaded by `n_threads` worker threads (<= 0: one per usable
 * hardware thread, at most 64; an explicit count is honoured up to 256).  keep > 0: reads longer than
 * 2*keep contribute their first and last `keep` samples only (windows are cut from those:
 * reference classify.py:337-349); keep <= 0: whole reads.  Read i occupies
 * samples[offsets[i] .. offsets[i+1]); a file that could not be read has status != F5_OK, an
 * empty id and an empty range (the reference skips such files, load_fast5s.py:47-49).  The
 * result owns its memory until f5_batch_free. */
typedef struct f5_batch f5_batch;
int f5_load_batch(const char* const* paths, int64_t n_files, int64_t keep, int n_threads,
                  f5_batch** out);
/* The same for reads [first, first + count) of ONE file - the way into multi-read fast5 files
 * (realtime.py:183-190 unpacks those with an external tool first): the file is opened and parsed
 * once, the worker threads share it and are dealt the reads one by one.  Layout of the result as
 * above, except that a read whose damage only shows while its Signal is inflated keeps its range,
 * zero filled: go by the status. */
int f5_load_reads(const char* path, int64_t first, int64_t count, int64_t keep, int n_threads,
                  f5_batch** out);                        /* count < 0: from `first` to the end */
/* Multi-read containers as a STREAM (BASELINE.json configs[4]; the reference's flow is
 * realtime.py:81-108 + :183-190, one file at a time through multi_to_single_fast5): the containers
 * of `paths` are loaded by a team of n_threads threads (<= 0: one per hardware thread, at most
 * 64; explicit counts up to 256) working on `depth` containers at once (<= 0: 3) - opening and
 * walking container k+1, k+2 beside the inflating of container k - and handed out in path order.
 * f5_stream_next blocks until the next container is ready: *index = its position in `paths`,
 * *container_status = F5_OK or why it could not be opened (then *batch is NULL); the batch is laid
 * out as f5_load_reads' and is the caller's to free.  Returns F5_ERR_NO_READ after the last one.
 * One consumer thread per stream. */
typedef struct f5_stream f5_stream;
int f5_stream_open(const char* const* paths, int64_t n_paths, int64_t keep, int n_threads,
                   int depth, f5_stream** out);
int f5_stream_next(f5_stream* stream, int64_t* index, int* container_status, f5_batch** batch);
void f5_stream_close(f5_stream* stream);

/* The same stream handing out the Signal AS STORED, for a decoder elsewhere - the GPU
 * (deepbinner_hip.h: dbh_inflate_dev, dbh_classify_pair_deflated).  Inflating is ~85 % of what
 * loading a read costs a CPU core; a raw batch costs the host the parsing and one pread per chunk.
 * A batch then holds no samples (f5_batch_samples is NULL) but
 *   - offsets (n_reads + 1, in SAMPLES): where each read's signal will lie once decoded;
 *   - a byte buffer (f5_batch_comp, f5_batch_comp_bytes; readable for 64 bytes beyond its end);
 *   - one f5_raw_stream per piece of stored Signal (a chunk, or a whole unchunked dataset), longest
 *     deflate stream first: where its bytes lie in the byte buffer, where its output goes in the
 *     sample buffer (byte offsets) and how many bytes of it are wanted, and whether it is a zlib
 *     stream (F5_RAW_ZLIB) or the bytes themselves (F5_RAW_STORED: unfiltered data, chunks with
 *     filters beyond deflate / fletcher32 - inflated and unshuffled by the host after all - and
 *     the deflate streams the host is to keep: those longer than host_inflate_above bytes, or,
 *     with host_inflate_above = -p (1..100), the longest ones of each container holding p per
 *     cent of its compressed bytes - none in a container whose deflate streams average more than
 *     64 KiB: long reads throughout are the GPU's alone; 0: none).  `reserved` = the
 *     index of the read the piece belongs to.  Layout identical to dbh_inflate_stream.
 * depth <= 0: half the team, between 3 and 8 (a raw container is ~30 ms of CPU behind a serial
 * 5-6 ms of parsing: three in flight starve sixteen threads). */
#define F5_RAW_ZLIB 0
#define F5_RAW_STORED 1
typedef struct f5_raw_stream {
    int64_t comp_offset, comp_bytes;
    int64_t out_offset, out_bytes;
    int32_t mode, reserved;
} f5_raw_stream;
int f5_stream_open_raw(const char* const* paths, int64_t n_paths, int n_threads, int depth,
                       int64_t host_inflate_above, f5_stream** out);
/* the same for a batch of one-read files (f5_load_batch's twin): one batch, read i = file i */
int f5_load_batch_raw(const char* const* paths, int64_t n_files, int n_threads,
                      int64_t host_inflate_above, f5_batch** out);
const uint8_t* f5_batch_comp(const f5_batch* batch);
int64_t f5_batch_comp_bytes(const f5_batch* batch);
const f5_raw_stream* f5_batch_streams(const f5_batch* batch);
int64_t f5_batch_n_streams(const f5_batch* batch);

/* The reads of a multi-read container as one-read fast5 files - what `deepbinner realtime` files
 * into the directories of their barcodes (the reference runs ont_fast5_api's
 * multi_to_single_fast5 and moves its output: realtime.py:183-190, :111-150).  Read
 * read_index[i] of `container` is written to out_paths[i]: /read_<id>/Raw/Signal with the
 * attributes of the read's group, of Raw, and its channel_id / tracking_id / context_tags groups
 * (scalar strings, integers, floats; what a basecaller needs), in the layout of
 * deepbinner_amd/hdf5_write.py (superblock 0, symbol-table groups, one deflated chunk), byte for
 * byte.  A Signal stored as ONE deflate-compressed chunk - what MinKNOW and ont_fast5_api write -
 * is carried over as stored: nothing is inflated, nothing deflated again.  A file is written
 * under a temporary name beside its own and linked into place: an existing file is never
 * overwritten (status F5_ERR_EXISTS), no symlink followed, no partial file left under the final
 * name.  status[i] = F5_OK or why read i could not be written; n_threads as everywhere.  *bytes_written (may be NULL): total
 * size of the files. */
int f5_write_single_reads(const char* container, int64_t n, const int64_t* read_index,
                          const char* const* out_paths, int n_threads, int32_t* status,
                          int64_t* bytes_written);
/* the bytes such a file would hold (tests); *size is set even if capacity is too small */
int f5_single_read_image(const char* container, int64_t read_index, uint8_t* out, int64_t capacity,
                         int64_t* size);

/* Where the packed samples of a batch live.  Freed batches leave their sample buffer in a pool
 * (bounded by DEEPBINNER_FAST5_POOL_MB, default 2048) for the next batch that fits, so that a
 * steady stream of containers allocates - and page-faults - nothing.  A caller may supply the
 * memory: alloc(bytes, user) / release(ptr, user), e.g. dbh_host_alloc / dbh_host_release of
 * libdeepbinner_hip.so (pinned host memory: the GPU's DMA engine then reads a batch where the
 * loader threads wrote it, deepbinner_hip.h).  NULL, NULL = malloc.  Batches alive at the time of
 * the call keep (and later return) the memory they have. */
typedef void* (*f5_alloc_fn)(size_t bytes, void* user);
typedef void (*f5_free_fn)(void* ptr, void* user);
int f5_set_sample_allocator(f5_alloc_fn alloc, f5_free_fn release, void* user);
void f5_release_idle_buffers(void);
int64_t f5_batch_size(const f5_batch* batch);             /* files / reads in the batch */
const int16_t* f5_batch_samples(const f5_batch* batch);
const int64_t* f5_batch_offsets(const f5_batch* batch);   /* n_files + 1 */
const int32_t* f5_batch_status(const f5_batch* batch);    /* n_files */
const char* f5_batch_read_ids(const f5_batch* batch);     /* n_files x F5_READ_ID_MAX */
void f5_batch_free(f5_batch* batch);

#ifdef __cplusplus
}
#endif
#endif
