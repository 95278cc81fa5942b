// libdeepbinner_fast5.so - native fast5 loader (C ABI: include/deepbinner_fast5.h).
//
// The reference reads fast5 files through h5py (deepbinner/load_fast5s.py:19,25-49); this is the
// slice of the HDF5 file format those files use, restated in C++ from the format specification
// exactly as deepbinner_amd/hdf5_lite.py restates it in Python (that module stays the readable
// description and the parity reference for this one: tests/test_fast5_native.py).
// Host-only: g++ -O2 -shared -fPIC fast5_reader.cpp -lz -ldl -pthread (libdeflate, if the system has
// it, is looked up at run time).
#include "../../include/deepbinner_fast5.h"

#include <dlfcn.h>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/syscall.h>
#include <sys/uio.h>
#include <linux/fs.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <pthread.h>
#include <deque>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <climits>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace {

constexpr uint64_t kUndef = ~0ull;
constexpr int kO = 8, kL = 8;          // only 8-byte offsets / lengths are supported
constexpr int kMaxDepth = 32;          // B-tree / indirect-block recursion guard

struct FormatError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// a compression filter this reader does not implement (ONT's VBZ, id 32020, is the one in the
// field): reported apart from damage, so that the caller can say what is the matter
struct UnsupportedFilter : FormatError {
    using FormatError::FormatError;
};

// the writer found a file under the name it was to create (nothing is ever overwritten)
struct ExistsError : std::runtime_error {
    ExistsError() : std::runtime_error("file exists") {}
};

// errno values with which link() says "this filesystem has no hard links": ENOTSUP / EOPNOTSUPP /
// ENOSYS, and EPERM, which is what FAT / exFAT and several FUSE and SMB mounts answer.  (Not EACCES,
// EXDEV or EMLINK: those are errors of this call, not properties of the filesystem, and fall
// through to "cannot name file".)
static bool no_hard_links(int e) {
    return e == EPERM || e == ENOTSUP || e == EOPNOTSUPP || e == ENOSYS;
}
// DEEPBINNER_FAST5_NO_LINK=1 (tests): link() is not tried at all, as on such a filesystem
static bool links_forbidden() {
    static const bool forced = [] {
        const char* v = std::getenv("DEEPBINNER_FAST5_NO_LINK");
        return v && v[0] == '1';
    }();
    return forced;
}
// The image under a name that must not exist yet (O_EXCL; a symlink there is not followed); a
// failed write leaves nothing behind.
static void write_exclusive(const char* path, const std::string& image) {
    const int fd = ::open(path, O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW | O_CLOEXEC, 0666);
    if (fd < 0) {
        if (errno == EEXIST) throw ExistsError();
        throw std::runtime_error("cannot create file");
    }
    size_t done = 0;
    while (done < image.size()) {
        const ssize_t k = ::write(fd, image.data() + done, image.size() - done);
        if (k <= 0) {
            ::close(fd);
            ::unlink(path);
            throw std::runtime_error("cannot write file");
        }
        done += (size_t)k;
    }
    if (::close(fd) != 0) {
        ::unlink(path);
        throw std::runtime_error("cannot close file");
    }
}

struct Msg {
    uint32_t type;
    uint64_t off;
    uint32_t size;
};

inline uint64_t pad8(uint64_t n) { return (n + 7) & ~7ull; }

inline int bit_length(uint64_t v) {
    int n = 0;
    while (v) {
        ++n;
        v >>= 1;
    }
    return n;
}

struct Filter {
    int id;
    std::vector<uint32_t> cd;
};

struct SignalInfo {
    int64_t n = 0;             // samples
    int layout = -1;           // 0 compact, 1 contiguous, 2 chunked
    uint64_t addr = kUndef;    // contiguous data / chunk B-tree
    uint64_t compact_off = 0, compact_size = 0;
    int64_t chunk_elems = 0;
    // chunk index: 0 = version-1 B-tree (layout message up to version 3); layout version 4
    // (HDF5 1.10, libver latest): 1 = the single chunk itself, 2 = implicit (all chunks back to
    // back), 3 = fixed array, 4 = extensible array
    int index = 0;
    uint64_t single_bytes = 0;
    uint32_t single_mask = 0;
    bool unfiltered_edge = false;   // layout flag: a partial last chunk is stored unfiltered
    std::vector<Filter> filters;
};

// libdeflate, when the system has it (looked up at run time: the image ships libdeflate.so.0 but
// no header): its whole-buffer inflate is 2-3 times as fast as zlib's streaming one, and inflating
// is most of what loading a read costs.  Same input (RFC 1950 streams, which is what the HDF5
// deflate filter writes), same output; anything it does not like - including an output that turns
// out larger than the chunk size promised - goes to zlib, which also decides what counts as
// corrupt.  DEEPBINNER_FAST5_INFLATE=zlib keeps it out.
struct LibDeflate {
    using Alloc = void* (*)();
    using Free = void (*)(void*);
    using Inflate = int (*)(void*, const void*, size_t, void*, size_t, size_t*);
    Alloc alloc = nullptr;
    Free release = nullptr;
    Inflate zlib_decompress = nullptr;
    LibDeflate() {
        void* lib = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        if (!lib) return;
        alloc = reinterpret_cast<Alloc>(dlsym(lib, "libdeflate_alloc_decompressor"));
        release = reinterpret_cast<Free>(dlsym(lib, "libdeflate_free_decompressor"));
        zlib_decompress = reinterpret_cast<Inflate>(dlsym(lib, "libdeflate_zlib_decompress"));
        if (!alloc || !release || !zlib_decompress) alloc = nullptr;
    }
    bool usable() const {
        if (!alloc) return false;
        const char* choice = std::getenv("DEEPBINNER_FAST5_INFLATE");
        return !(choice && std::strcmp(choice, "zlib") == 0);
    }
};
const LibDeflate& libdeflate() {
    static const LibDeflate lib;
    return lib;
}

// The chunk inflated last.  A read stored as ONE chunk (common) is asked for twice, once per end,
// and deflate cannot be entered in the middle.  One per thread: a Fast5 whose reads are resolved
// is read-only otherwise, so several threads can decode different reads of it at once.
// It also owns what decoding needs over and over - the second buffer of the filter pipeline and
// the inflate state - so that a thread working through thousands of reads does not allocate and
// free ~150 KB per read (in a fresh malloc arena that is a heap grow / trim per read, which
// serialises the threads on the address-space lock).
struct ChunkCache {
    std::vector<uint8_t> data, scratch;
    uint64_t addr = ~0ull, bytes = 0;
    const void* owner = nullptr;        // the file the cached chunk came from (addresses repeat
                                        // from file to file: a thread may serve several)
    z_stream zs;
    bool zs_ready = false;
    void* fast_inflater = nullptr;      // libdeflate's, when there is one
    ChunkCache() { std::memset(&zs, 0, sizeof(zs)); }
    ~ChunkCache() {
        if (zs_ready) inflateEnd(&zs);
        if (fast_inflater) libdeflate().release(fast_inflater);
    }
    ChunkCache(const ChunkCache&) = delete;
    ChunkCache& operator=(const ChunkCache&) = delete;
};

struct ReadEntry {
    uint64_t group_addr = 0;          // the group that holds Signal and read_id (".../Raw")
    uint64_t read_group_addr = 0;     // multi-read containers: /read_<id> itself (0 otherwise)
    bool resolved = false;
    std::string read_id;
    SignalInfo signal;
};

class Fast5 {
  public:
    // `reuse`: a buffer of the caller's (a worker thread's) that small files are read into
    // instead of one allocated per file - a few hundred KB per file is above malloc's mmap
    // threshold, i.e. an mmap + munmap per file, which many threads queue up on
    explicit Fast5(const char* path, std::vector<uint8_t>* reuse = nullptr) {
        fd_ = ::open(path, O_RDONLY);
        if (fd_ < 0) throw std::runtime_error("cannot open file");
        struct stat st;
        if (fstat(fd_, &st) != 0 || st.st_size <= 0) {
            ::close(fd_);
            fd_ = -1;
            throw std::runtime_error("cannot stat file / file is empty");
        }
        len_ = (uint64_t)st.st_size;
        if (len_ <= kReadWhole) {
            // one-read files are a few hundred KB: read() them - with many loader threads every
            // mmap/munmap would queue on the process-wide address-space lock
            std::vector<uint8_t>& bytes = reuse ? *reuse : owned_;
            if (bytes.size() < len_) bytes.resize((size_t)len_);
            uint64_t got = 0;
            while (got < len_) {
                const ssize_t k = ::pread(fd_, bytes.data() + got, (size_t)(len_ - got), (off_t)got);
                if (k <= 0) {
                    ::close(fd_);
                    fd_ = -1;
                    throw std::runtime_error("cannot read file");
                }
                got += (uint64_t)k;
            }
            ::close(fd_);
            fd_ = -1;
            buf_ = bytes.data();
            return;
        }
        void* p = mmap(nullptr, len_, PROT_READ, MAP_PRIVATE, fd_, 0);
        if (p == MAP_FAILED) {
            ::close(fd_);
            fd_ = -1;
            throw std::runtime_error("cannot map file");
        }
        buf_ = static_cast<const uint8_t*>(p);
        mapped_ = true;
    }
    ~Fast5() {
        if (mapped_) munmap(const_cast<uint8_t*>(buf_), len_);
        if (fd_ >= 0) ::close(fd_);
    }
    Fast5(const Fast5&) = delete;
    Fast5& operator=(const Fast5&) = delete;

    // ---- what the C ABI needs ----------------------------------------------------------------
    void parse() {
        parse_superblock();
        find_reads();
    }
    int layout() const { return layout_; }
    int64_t n_reads() const { return (int64_t)reads_.size(); }

    const ReadEntry& read(int64_t index) {
        if (index < 0 || index >= (int64_t)reads_.size()) throw std::out_of_range("read index");
        ReadEntry& r = reads_[(size_t)index];
        if (!r.resolved) {
            const std::vector<Msg> msgs = object_header(r.group_addr);
            if (!find_string_attribute(msgs, "read_id", &r.read_id))
                throw std::out_of_range("read group without read_id");
            const std::map<std::string, uint64_t> links = group_links(msgs);
            auto it = links.find("Signal");
            if (it == links.end()) throw std::out_of_range("read group without Signal");
            r.signal = signal_info(it->second);
            r.resolved = true;
        }
        return r;
    }

    void read_signal(const SignalInfo& s, int64_t first, int64_t count, int16_t* out,
                     ChunkCache* cache = nullptr) {
        if (!cache) cache = &cache_;
        if (first < 0 || count < 0 || first + count > s.n) throw std::out_of_range("sample range");
        if (count == 0) return;
        if (s.layout == 0) {
            if ((uint64_t)(first + count) * 2 > s.compact_size) throw FormatError("compact data too short");
            need(s.compact_off, s.compact_size);
            std::memcpy(out, buf_ + s.compact_off + first * 2, (size_t)count * 2);
        } else if (s.layout == 1) {
            if (s.addr == kUndef) {
                std::memset(out, 0, (size_t)count * 2);
                return;
            }
            const uint64_t start = base_ + s.addr;
            need(start, (uint64_t)s.n * 2);
            std::memcpy(out, buf_ + start + first * 2, (size_t)count * 2);
        } else {
            std::memset(out, 0, (size_t)count * 2);
            if (s.addr == kUndef) return;
            if (s.chunk_elems <= 0) throw FormatError("bad chunk size");
            if (s.index == 0) {
                walk_chunks(s, s.addr, first, count, out, 0, cache);
                return;
            }
            for (int64_t k = first / s.chunk_elems; k * s.chunk_elems < first + count; ++k) {
                uint64_t addr = kUndef, nbytes = 0;
                uint32_t mask = 0;
                if (!indexed_chunk(s, (uint64_t)k, &addr, &nbytes, &mask)) continue;
                const bool partial = (k + 1) * s.chunk_elems > s.n;
                if (s.unfiltered_edge && partial) mask = ~0u;
                copy_from_chunk(s, k * s.chunk_elems, addr, nbytes, mask, first, count, out, cache);
            }
        }
    }

    // ---- the Signal as it is stored (for callers that inflate elsewhere: the GPU) ---------------
    // What a read's Signal consists of on disk, piece by piece, in sample order.  A piece covers
    // samples [first, first + count) of the read.
    struct RawPiece {
        int kind = 0;              // kZlib / kStored / kHostDecode / kZeros
        uint64_t file_off = 0;     // where its bytes lie in the file (kZlib, kStored)
        uint64_t nbytes = 0;       // how many
        int64_t first = 0, count = 0;
        uint64_t chunk_addr = 0, chunk_bytes = 0;      // kHostDecode: the chunk, for decode_chunk
        uint32_t mask = 0;
        int64_t comp_offset = 0, comp_bytes = 0;       // its place in a batch's byte buffer
    };
    enum { kZlib = 0, kStored = 1, kHostDecode = 2, kZeros = 3 };

    // zlib_above: deflate streams longer than this many bytes are left to the host (a lane of the
    // GPU decoder walks ONE stream: a stream ten times the usual length holds its wave ten times
    // as long, while a CPU core inflates it in a millisecond); <= 0: no limit.
    void signal_pieces(const SignalInfo& s, int64_t zlib_above, std::vector<RawPiece>* out) const {
        out->clear();
        if (s.n <= 0) return;
        RawPiece p;
        p.first = 0;
        p.count = s.n;
        if (s.layout == 0) {
            if ((uint64_t)s.n * 2 > s.compact_size) throw FormatError("compact data too short");
            need(s.compact_off, s.compact_size);
            p.kind = kStored;
            p.file_off = s.compact_off;
            p.nbytes = (uint64_t)s.n * 2;
            out->push_back(p);
            return;
        }
        if (s.layout == 1) {
            if (s.addr == kUndef) {
                p.kind = kZeros;
            } else {
                p.kind = kStored;
                p.file_off = base_ + s.addr;
                p.nbytes = (uint64_t)s.n * 2;
                need(p.file_off, p.nbytes);
            }
            out->push_back(p);
            return;
        }
        if (s.chunk_elems <= 0) throw FormatError("bad chunk size");
        const int64_t n_chunks = (s.n + s.chunk_elems - 1) / s.chunk_elems;
        if (n_chunks > (1 << 24)) throw FormatError("implausible number of chunks");
        std::vector<RawPiece> by_chunk((size_t)n_chunks);
        for (int64_t k = 0; k < n_chunks; ++k) {
            RawPiece& q = by_chunk[(size_t)k];
            q.kind = kZeros;                       // a chunk that was never written reads as zeros
            q.first = k * s.chunk_elems;
            q.count = std::min<int64_t>(s.chunk_elems, s.n - q.first);
        }
        auto found = [&](int64_t k, uint64_t addr, uint64_t nbytes, uint32_t mask) {
            if (k < 0 || k >= n_chunks) return;
            RawPiece& q = by_chunk[(size_t)k];
            const uint64_t start = file_off(addr);
            need(start, nbytes);
            q.chunk_addr = addr;
            q.chunk_bytes = nbytes;
            q.mask = mask;
            // the filters that were applied to THIS chunk, in the order they were applied
            int applied[8], n_applied = 0;
            bool known = true;
            for (size_t i = 0; i < s.filters.size(); ++i) {
                if (mask & (1u << i)) continue;
                if (n_applied == 8) known = false;
                else applied[n_applied++] = s.filters[i].id;
            }
            q.file_off = start;
            q.nbytes = nbytes;
            if (!known) {
                q.kind = kHostDecode;
            } else if (n_applied == 0) {
                q.kind = kStored;
            } else if (n_applied == 1 && applied[0] == 1) {
                q.kind = kZlib;
            } else if (n_applied == 2 && applied[0] == 1 && applied[1] == 3 && nbytes >= 4) {
                q.kind = kZlib;                    // deflate, then a checksum behind the stream
                q.nbytes = nbytes - 4;
            } else if (n_applied == 2 && applied[0] == 3 && applied[1] == 1) {
                q.kind = kZlib;                    // the checksum lies inside, behind the samples
            } else if (n_applied == 1 && applied[0] == 3 && nbytes >= 4) {
                q.kind = kStored;
                q.nbytes = nbytes - 4;
            } else {
                q.kind = kHostDecode;              // shuffle, or an order not seen in the field
            }
            if (q.kind == kZlib && zlib_above > 0 && (int64_t)q.nbytes > zlib_above)
                q.kind = kHostDecode;
        };
        if (s.addr != kUndef) {
            if (s.index == 0) {
                collect_chunks(s, s.addr, 0, found);
            } else {
                for (int64_t k = 0; k < n_chunks; ++k) {
                    uint64_t addr = kUndef, nbytes = 0;
                    uint32_t mask = 0;
                    if (!indexed_chunk(s, (uint64_t)k, &addr, &nbytes, &mask)) continue;
                    if (s.unfiltered_edge && (k + 1) * s.chunk_elems > s.n) mask = ~0u;
                    found(k, addr, nbytes, mask);
                }
            }
        }
        *out = std::move(by_chunk);
    }

    // bytes [off, off + n) of the file (large files: pread, not the mapping - see decode_chunk)
    void read_bytes(uint64_t off, uint64_t n, uint8_t* dst) const {
        need(off, n);
        if (mapped_ && fd_ >= 0) {
            uint64_t got = 0;
            while (got < n) {
                const ssize_t k = ::pread(fd_, dst + got, (size_t)(n - got), (off_t)(off + got));
                if (k <= 0) throw FormatError("cannot read chunk");
                got += (uint64_t)k;
            }
        } else {
            std::memcpy(dst, buf_ + off, (size_t)n);
        }
    }

    // MANY byte ranges of the file at once (the stored Signal pieces of a stretch of a multi-read
    // container, f5_stream's raw batches): sorted by their place in the file, neighbours with small
    // gaps between them - the object headers and B-tree nodes between two reads' chunks - are read
    // by ONE preadv (the gaps go to a sink), up to kMaxIov ranges a call, where a pread per range
    // was a system call per read (half of the raw loader's time).  Ranges are checked against the
    // file's length first; failed[k] = 1 for every range that could not be read.
    struct IoItem {
        uint64_t off, n;
        uint8_t* dst;
        int64_t tag;                 // (the caller's: which read it belongs to)
    };
    void read_many(std::vector<IoItem>& items, std::vector<char>* failed) const {
        failed->assign(items.size(), 0);
        std::vector<size_t> order;
        order.reserve(items.size());
        for (size_t k = 0; k < items.size(); ++k) {
            const IoItem& it = items[k];
            if (it.off > len_ || it.n > len_ - it.off) (*failed)[k] = 1;
            else if (it.n > 0) order.push_back(k);
        }
        if (!(mapped_ && fd_ >= 0)) {
            for (size_t k : order) std::memcpy(items[k].dst, buf_ + items[k].off, (size_t)items[k].n);
            return;
        }
        std::sort(order.begin(), order.end(),
                  [&](size_t x, size_t y) { return items[x].off < items[y].off; });
        constexpr uint64_t kMaxGap = 32u << 10;
        constexpr size_t kMaxIov = 512;
        thread_local std::vector<uint8_t> sink(kMaxGap);
        thread_local std::vector<struct iovec> iov;
        auto one_by_one = [&](size_t a, size_t b) {
            for (size_t j = a; j < b; ++j) {
                const IoItem& it = items[order[j]];
                uint64_t got = 0;
                while (got < it.n) {
                    const ssize_t k = ::pread(fd_, it.dst + got, (size_t)(it.n - got), (off_t)(it.off + got));
                    if (k <= 0) {
                        (*failed)[order[j]] = 1;
                        break;
                    }
                    got += (uint64_t)k;
                }
            }
        };
        size_t a = 0;
        while (a < order.size()) {
            iov.clear();
            size_t b = a;
            uint64_t pos = items[order[a]].off;
            while (b < order.size() && iov.size() + 2 <= kMaxIov) {
                const IoItem& it = items[order[b]];
                if (it.off < pos) break;                      // (overlapping ranges: on their own)
                const uint64_t gap = it.off - pos;
                if (b > a && gap > kMaxGap) break;
                if (gap > 0) iov.push_back({sink.data(), (size_t)gap});
                iov.push_back({it.dst, (size_t)it.n});
                pos = it.off + it.n;
                ++b;
            }
            if (b == a) {                                      // (an overlap at the run's start)
                one_by_one(a, a + 1);
                a += 1;
                continue;
            }
            const uint64_t total = pos - items[order[a]].off;
            const ssize_t k = b - a > 1 ? ::preadv(fd_, iov.data(), (int)iov.size(), (off_t)items[order[a]].off) : -1;
            if (k < 0 || (uint64_t)k != total) one_by_one(a, b);       // (short or failed: range by range)
            a = b;
        }
    }

    // a piece the host decodes itself -> its samples' bytes at dst (count * 2 of them)
    void decode_piece(const SignalInfo& s, const RawPiece& p, uint8_t* dst, ChunkCache* cache) const {
        decode_chunk(s, p.chunk_addr, p.chunk_bytes, p.mask, cache);
        cache->owner = nullptr;                    // (not a chunk read_signal may reuse blindly)
        std::memcpy(dst, cache->data.data(), (size_t)p.count * 2);
    }

  private:
    static constexpr uint64_t kReadWhole = 8u << 20;   // files up to 8 MiB are read, larger mapped
    int fd_ = -1;
    bool mapped_ = false;
    std::vector<uint8_t> owned_;
    const uint8_t* buf_ = nullptr;
    uint64_t len_ = 0;
    uint64_t base_ = 0, root_addr_ = 0;
    int layout_ = F5_LAYOUT_NONE;
    std::vector<ReadEntry> reads_;
    ChunkCache cache_;                           // for callers that bring none (one thread)

    // ---- checked access to the mapped file ---------------------------------------------------
    void need(uint64_t off, uint64_t n) const {
        if (off > len_ || n > len_ - off) throw FormatError("offset beyond end of file");
    }
    uint64_t u(uint64_t off, int size) const {
        need(off, (uint64_t)size);
        uint64_t v = 0;
        for (int i = size - 1; i >= 0; --i) v = (v << 8) | buf_[off + i];
        return v;
    }
    uint8_t b(uint64_t off) const {
        need(off, 1);
        return buf_[off];
    }
    bool sig(uint64_t off, const char* s4) const {
        need(off, 4);
        return std::memcmp(buf_ + off, s4, 4) == 0;
    }
    uint64_t file_off(uint64_t addr) const {
        if (addr > len_ || base_ > len_ - addr) throw FormatError("address beyond end of file");
        return base_ + addr;
    }

    // ---- superblock (hdf5_lite._parse_superblock) ---------------------------------------------
    void parse_superblock() {
        static const uint8_t kSig[8] = {0x89, 'H', 'D', 'F', '\r', '\n', 0x1a, '\n'};
        uint64_t off = 0;
        while (true) {
            if (off + 8 > len_) throw FormatError("HDF5 signature not found");
            if (std::memcmp(buf_ + off, kSig, 8) == 0) break;
            off = off == 0 ? 512 : off * 2;
        }
        const int version = b(off + 8);
        int O, L;
        if (version == 0 || version == 1) {
            O = b(off + 13);
            L = b(off + 14);
            if (O != kO || L != kL) throw FormatError("only 8-byte offsets/lengths are supported");
            uint64_t p = off + 24 + (version == 1 ? 4 : 0);
            base_ = u(p, kO);
            p += 4 * kO;
            root_addr_ = u(p + kO, kO);
        } else if (version == 2 || version == 3) {
            O = b(off + 9);
            L = b(off + 10);
            if (O != kO || L != kL) throw FormatError("only 8-byte offsets/lengths are supported");
            base_ = u(off + 12, kO);
            root_addr_ = u(off + 12 + 3 * kO, kO);
        } else {
            throw FormatError("unsupported superblock version");
        }
        if (base_ == 0 && off) base_ = off;
    }

    // ---- object headers (hdf5_lite._read_object_header) ---------------------------------------
    std::vector<Msg> object_header(uint64_t addr) const {
        const uint64_t start = file_off(addr);
        need(start, 16);
        std::vector<Msg> messages;
        std::vector<std::pair<uint64_t, uint64_t>> blocks;
        if (sig(start, "OHDR")) {
            if (b(start + 4) != 2) throw FormatError("unsupported object header version");
            const int flags = b(start + 5);
            uint64_t p = start + 6;
            if (flags & 0x20) p += 16;
            if (flags & 0x10) p += 4;
            const int csize = 1 << (flags & 3);
            const uint64_t chunk0 = u(p, csize);
            p += csize;
            const bool track_order = flags & 0x04;
            blocks.emplace_back(p, chunk0);
            for (size_t i = 0; i < blocks.size(); ++i) {
                if (blocks.size() > 4096) throw FormatError("too many header continuation blocks");
                uint64_t bp = blocks[i].first;
                const uint64_t end = bp + blocks[i].second;
                need(bp, blocks[i].second);
                while (bp + 4 <= end) {
                    const uint32_t mtype = b(bp);
                    const uint32_t msize = (uint32_t)u(bp + 1, 2);
                    bp += 4 + (track_order ? 2 : 0);
                    if (mtype == 0x10) {
                        const uint64_t caddr = u(bp, kO), clen = u(bp + kO, kL);
                        if (clen < 8) throw FormatError("bad continuation block");
                        blocks.emplace_back(file_off(caddr) + 4, clen - 8);
                    } else if (mtype != 0) {
                        messages.push_back(Msg{mtype, bp, msize});
                    }
                    bp += msize;
                }
            }
            return messages;
        }
        if (b(start) != 1) throw FormatError("unsupported object header version");
        const uint64_t nmsgs = u(start + 2, 2);
        const uint64_t hsize = u(start + 8, 4);
        blocks.emplace_back(start + 16, hsize);
        uint64_t seen = 0;
        for (size_t i = 0; i < blocks.size() && seen < nmsgs; ++i) {
            uint64_t bp = blocks[i].first;
            const uint64_t end = bp + blocks[i].second;
            need(bp, blocks[i].second);
            while (bp + 8 <= end && seen < nmsgs) {
                const uint32_t mtype = (uint32_t)u(bp, 2);
                const uint32_t msize = (uint32_t)u(bp + 2, 2);
                bp += 8;
                ++seen;
                if (mtype == 0x0010) {
                    const uint64_t caddr = u(bp, kO), clen = u(bp + kO, kL);
                    blocks.emplace_back(file_off(caddr), clen);
                } else if (mtype != 0) {
                    messages.push_back(Msg{mtype, bp, msize});
                }
                bp += msize;
            }
        }
        return messages;
    }

    static const Msg* first_of(const std::vector<Msg>& msgs, uint32_t type) {
        for (const Msg& m : msgs)
            if (m.type == type) return &m;
        return nullptr;
    }

    // ---- groups (hdf5_lite.Group._load_links) --------------------------------------------------
    std::string cstring(uint64_t off) const {
        need(off, 1);
        const void* end = std::memchr(buf_ + off, 0, (size_t)(len_ - off));
        if (!end) throw FormatError("unterminated string");
        return std::string(reinterpret_cast<const char*>(buf_ + off),
                           (size_t)(static_cast<const uint8_t*>(end) - (buf_ + off)));
    }

    void walk_group_btree(uint64_t addr, uint64_t heap_data, std::map<std::string, uint64_t>* links,
                          int depth) const {
        if (depth > kMaxDepth) throw FormatError("group B-tree too deep");
        const uint64_t p = file_off(addr);
        if (!sig(p, "TREE")) throw FormatError("bad group B-tree signature");
        const int level = b(p + 5);
        const uint64_t used = u(p + 6, 2);
        const uint64_t q = p + 8 + 2 * kO;
        for (uint64_t i = 0; i < used; ++i) {
            const uint64_t child = u(q + kL + i * (kL + kO), kO);
            if (level > 0) {
                walk_group_btree(child, heap_data, links, depth + 1);
            } else {
                const uint64_t s = file_off(child);
                if (!sig(s, "SNOD")) throw FormatError("bad symbol table node signature");
                const uint64_t nsym = u(s + 6, 2);
                uint64_t e = s + 8;
                for (uint64_t k = 0; k < nsym; ++k) {
                    const uint64_t name_off = u(e, kO), ohdr = u(e + kO, kO);
                    (*links)[cstring(heap_data + name_off)] = ohdr;
                    e += 2 * kO + 24;
                }
            }
        }
    }

    // link message -> (name, object header address); false for soft/external links
    bool parse_link(uint64_t off, std::string* name, uint64_t* addr) const {
        if (b(off) != 1) throw FormatError("bad link message version");
        const int flags = b(off + 1);
        uint64_t p = off + 2;
        int ltype = 0;
        if (flags & 0x08) ltype = b(p++);
        if (flags & 0x04) p += 8;
        if (flags & 0x10) p += 1;
        const int lsize = 1 << (flags & 3);
        const uint64_t nlen = u(p, lsize);
        p += lsize;
        need(p, nlen);
        name->assign(reinterpret_cast<const char*>(buf_ + p), (size_t)nlen);
        p += nlen;
        if (ltype != 0) return false;
        *addr = u(p, kO);
        return true;
    }

    std::map<std::string, uint64_t> group_links(const std::vector<Msg>& msgs) const {
        std::map<std::string, uint64_t> links;
        if (const Msg* stab = first_of(msgs, 0x0011)) {
            const uint64_t btree = u(stab->off, kO), heap = u(stab->off + kO, kO);
            const uint64_t hp = file_off(heap);
            if (!sig(hp, "HEAP")) throw FormatError("bad local heap signature");
            const uint64_t heap_data = file_off(u(hp + 8 + 2 * kL, kO));
            walk_group_btree(btree, heap_data, &links, 0);
        }
        for (const Msg& m : msgs) {
            if (m.type != 0x0006) continue;
            std::string name;
            uint64_t addr;
            if (parse_link(m.off, &name, &addr)) links[name] = addr;
        }
        if (const Msg* linfo = first_of(msgs, 0x0002)) {
            const int flags = b(linfo->off + 1);
            const uint64_t p = linfo->off + 2 + ((flags & 1) ? 8 : 0);
            const uint64_t heap_addr = u(p, kO), index_addr = u(p + kO, kO);
            if (heap_addr != kUndef && index_addr != kUndef) {
                for (uint64_t obj_off : dense_objects(heap_addr, index_addr)) {
                    std::string name;
                    uint64_t addr;
                    if (parse_link(obj_off, &name, &addr)) links[name] = addr;
                }
            }
        }
        return links;
    }

    // ---- fractal heap + v2 B-tree (dense link / attribute storage) ----------------------------
    struct FractalHeap {
        uint64_t id_len = 0, max_managed = 0, width = 0, start_size = 0, max_direct = 0;
        uint64_t root_addr = kUndef, cur_rows = 0;
        int off_bytes = 0, max_direct_rows = 2;
    };

    FractalHeap fractal_heap(uint64_t addr) const {
        FractalHeap h;
        const uint64_t p = file_off(addr);
        if (!sig(p, "FRHP")) throw FormatError("bad fractal heap signature");
        h.id_len = u(p + 5, 2);
        if (u(p + 7, 2)) throw FormatError("filtered fractal heaps are not supported");
        h.max_managed = u(p + 10, 4);
        const uint64_t q = p + 14 + kL + kO + kL + kO + 8 * kL;
        h.width = u(q, 2);
        h.start_size = u(q + 2, kL);
        h.max_direct = u(q + 2 + kL, kL);
        const uint64_t max_heap_bits = u(q + 2 + 2 * kL, 2);
        h.root_addr = u(q + 6 + 2 * kL, kO);
        h.cur_rows = u(q + 6 + 2 * kL + kO, 2);
        h.off_bytes = (int)((max_heap_bits + 7) / 8);
        if (h.width == 0 || h.start_size == 0 || h.off_bytes > 8) throw FormatError("bad fractal heap");
        uint64_t size = h.start_size;
        while (size < h.max_direct && h.max_direct_rows < 64) {
            size <<= 1;
            ++h.max_direct_rows;
        }
        return h;
    }

    uint64_t heap_locate(const FractalHeap& h, uint64_t offset) const {
        if (h.root_addr == kUndef) throw FormatError("empty fractal heap");
        if (h.cur_rows == 0) return file_off(h.root_addr) + offset;
        uint64_t iaddr = h.root_addr, rel = offset;
        for (int depth = 0; depth < kMaxDepth; ++depth) {
            const uint64_t blk = file_off(iaddr);
            if (!sig(blk, "FHIB")) throw FormatError("bad fractal heap indirect block signature");
            const uint64_t first = h.width * h.start_size;
            const int row = rel < first ? 0 : bit_length(rel / first);
            if (row > 62) throw FormatError("bad fractal heap offset");
            const uint64_t row_start = row == 0 ? 0 : (h.width * h.start_size) << (row - 1);
            const uint64_t bsize = row == 0 ? h.start_size : h.start_size << (row - 1);
            const uint64_t col = (rel - row_start) / bsize;
            const uint64_t entry = blk + 5 + kO + h.off_bytes + ((uint64_t)row * h.width + col) * kO;
            const uint64_t child = u(entry, kO);
            if (child == kUndef) throw FormatError("heap object in an unallocated block");
            rel -= row_start + col * bsize;
            if (row < h.max_direct_rows) return file_off(child) + rel;
            iaddr = child;
        }
        throw FormatError("fractal heap too deep");
    }

    static int enc_size(uint64_t limit) { return (bit_length(std::max<uint64_t>(limit, 1)) - 1) / 8 + 1; }

    void btree2_walk(uint64_t naddr, uint64_t nrec, int d, uint64_t rec_size, int nrec_size,
                     const std::vector<int>& cum_size, std::vector<uint64_t>* records,
                     int depth) const {
        if (depth > kMaxDepth) throw FormatError("v2 B-tree too deep");
        const uint64_t blk = file_off(naddr);
        if (d == 0) {
            if (!sig(blk, "BTLF")) throw FormatError("bad v2 B-tree leaf signature");
            for (uint64_t i = 0; i < nrec; ++i) {
                need(blk + 6 + i * rec_size, rec_size);
                records->push_back(blk + 6 + i * rec_size);
            }
            return;
        }
        if (!sig(blk, "BTIN")) throw FormatError("bad v2 B-tree internal node signature");
        const uint64_t recs = blk + 6;
        const uint64_t ptrs = recs + nrec * rec_size;
        const uint64_t ptr = kO + nrec_size + (d > 1 ? cum_size[(size_t)d - 1] : 0);
        for (uint64_t i = 0; i <= nrec; ++i) {
            const uint64_t e = ptrs + i * ptr;
            const uint64_t child = u(e, kO);
            const uint64_t child_nrec = u(e + kO, nrec_size);
            btree2_walk(child, child_nrec, d - 1, rec_size, nrec_size, cum_size, records, depth + 1);
            if (i < nrec) {
                need(recs + i * rec_size, rec_size);
                records->push_back(recs + i * rec_size);
            }
        }
    }

    // file offsets of the raw records of a version-2 B-tree, in key order
    std::vector<uint64_t> btree2_records(uint64_t addr, uint64_t* rec_size_out, int* type_out) const {
        std::vector<uint64_t> records;
        const uint64_t p = file_off(addr);
        if (!sig(p, "BTHD")) throw FormatError("bad v2 B-tree signature");
        *type_out = b(p + 5);
        const uint64_t node_size = u(p + 6, 4);
        const uint64_t rec_size = u(p + 10, 2);
        const int depth = (int)u(p + 12, 2);
        const uint64_t root = u(p + 16, kO);
        const uint64_t root_nrec = u(p + 16 + kO, 2);
        *rec_size_out = rec_size;
        if (root == kUndef || root_nrec == 0) return records;
        if (rec_size == 0 || node_size < 16 || depth > kMaxDepth) throw FormatError("bad v2 B-tree header");
        std::vector<uint64_t> cum_max;
        std::vector<int> cum_size;
        const uint64_t max0 = (node_size - 10) / rec_size;
        cum_max.push_back(max0);
        const int nrec_size = enc_size(max0);
        cum_size.push_back(enc_size(max0));
        for (int d = 1; d <= depth; ++d) {
            const uint64_t ptr = kO + nrec_size + (d > 1 ? cum_size[(size_t)d - 1] : 0);
            const uint64_t m = (node_size - (10 + ptr)) / (rec_size + ptr);
            cum_max.push_back((m + 1) * cum_max[(size_t)d - 1] + m);
            cum_size.push_back(enc_size(cum_max[(size_t)d]));
        }
        btree2_walk(root, root_nrec, depth, rec_size, nrec_size, cum_size, &records, 0);
        return records;
    }

    // file offsets of the messages held in dense (fractal heap + v2 B-tree) storage
    std::vector<uint64_t> dense_objects(uint64_t heap_addr, uint64_t index_addr) const {
        const FractalHeap heap = fractal_heap(heap_addr);
        uint64_t rec_size = 0;
        int btype = 0;
        std::vector<uint64_t> out;
        for (uint64_t rec : btree2_records(index_addr, &rec_size, &btype)) {
            // type 5 (link name) record: hash(4) + heap id; type 8 (attribute name): heap id first
            const uint64_t id = btype == 5 ? rec + 4 : rec;
            if ((btype == 5 ? 4 : 0) + heap.id_len > rec_size) throw FormatError("bad dense record");
            if (((b(id) >> 4) & 3) != 0) throw FormatError("only managed fractal-heap objects are supported");
            out.push_back(heap_locate(heap, u(id + 1, heap.off_bytes)));
        }
        return out;
    }

    // ---- attributes: only string values are needed (read_id) -----------------------------------
    std::string global_heap_object(uint64_t coll_addr, uint64_t index) const {
        const uint64_t p = file_off(coll_addr);
        if (!sig(p, "GCOL")) throw FormatError("bad global heap signature");
        const uint64_t size = u(p + 8, kL);
        uint64_t o = p + 8 + kL;
        const uint64_t end = p + size;
        while (o + 8 + kL <= end) {
            const uint64_t idx = u(o, 2);
            const uint64_t osize = u(o + 8, kL);
            if (idx == 0) break;
            if (idx == index) {
                need(o + 8 + kL, osize);
                return std::string(reinterpret_cast<const char*>(buf_ + o + 8 + kL), (size_t)osize);
            }
            o += 8 + kL + pad8(osize);
        }
        throw FormatError("global heap object not found");
    }

    // attribute message at `off`: its name; if it is a scalar string, its value
    bool parse_string_attribute(uint64_t off, std::string* name, std::string* value) const {
        const int version = b(off);
        const uint64_t nsz = u(off + 2, 2), tsz = u(off + 4, 2), ssz = u(off + 6, 2);
        uint64_t p = off + 8;
        if (version == 3)
            p += 1;
        else if (version != 1 && version != 2)
            throw FormatError("unsupported attribute version");
        const bool padded = version == 1;
        need(p, nsz);
        name->assign(reinterpret_cast<const char*>(buf_ + p), (size_t)nsz);
        name->resize(std::strlen(name->c_str()));
        p += padded ? pad8(nsz) : nsz;
        const uint64_t dt_off = p;
        p += padded ? pad8(tsz) : tsz;
        const uint64_t ds_off = p;
        p += padded ? pad8(ssz) : ssz;
        // dataspace: scalar or a single element
        const int ds_version = b(ds_off), rank = b(ds_off + 1);
        if (ds_version != 1 && ds_version != 2) return false;
        uint64_t count = 1;
        const uint64_t dims = ds_off + (ds_version == 1 ? 8 : 4);
        for (int i = 0; i < rank; ++i) count *= u(dims + (uint64_t)i * kL, kL);
        if (count != 1) return false;
        const int cls = b(dt_off) & 0x0F;
        const uint64_t bits = u(dt_off + 1, 3);
        const uint64_t size = u(dt_off + 4, 4);
        if (cls == 3) {            // fixed-length string, NUL padding stripped like h5py does
            need(p, size);
            value->assign(reinterpret_cast<const char*>(buf_ + p), (size_t)size);
            value->resize(std::strlen(value->c_str()));
            return true;
        }
        if (cls == 9 && (bits & 0x0F) == 1) {   // variable-length string in the global heap
            const uint64_t coll = u(p + 4, kO), idx = u(p + 4 + kO, 4);
            *value = coll ? global_heap_object(coll, idx) : std::string();
            return true;
        }
        return false;
    }

    bool find_string_attribute(const std::vector<Msg>& msgs, const char* wanted,
                               std::string* value) const {
        std::string name, v;
        for (const Msg& m : msgs) {
            if (m.type == 0x000C) {
                if (parse_string_attribute(m.off, &name, &v) && name == wanted) {
                    *value = v;
                    return true;
                }
            } else if (m.type == 0x0015) {   // dense attribute storage
                const int flags = b(m.off + 1);
                const uint64_t p = m.off + 2 + ((flags & 1) ? 2 : 0);
                const uint64_t heap_addr = u(p, kO), index_addr = u(p + kO, kO);
                if (heap_addr == kUndef || index_addr == kUndef) continue;
                for (uint64_t obj_off : dense_objects(heap_addr, index_addr))
                    if (parse_string_attribute(obj_off, &name, &v) && name == wanted) {
                        *value = v;
                        return true;
                    }
            }
        }
        return false;
    }

    // ---- every scalar attribute of an object, as a one-read copy of the read carries it --------
    // (what deepbinner_amd/hdf5_write.py's attribute_from_value keeps of what hdf5_lite reads:
    // strings, integers of 1/2/4/8 bytes, floats of 4/8 bytes; arrays, enums, references and
    // anything else are left behind)
  public:
    struct Attr {
        int kind = 0;              // 0 string, 1 integer, 2 float
        int size = 0;              // bytes of an integer / a float
        bool is_signed = false;
        std::string bytes;         // the text without its NUL / the value, little-endian
    };
    typedef std::map<std::string, Attr> Attrs;

    bool parse_attribute(uint64_t off, std::string* name, Attr* a) const {
        const int version = b(off);
        const uint64_t nsz = u(off + 2, 2), tsz = u(off + 4, 2), ssz = u(off + 6, 2);
        uint64_t p = off + 8;
        if (version == 3)
            p += 1;
        else if (version != 1 && version != 2)
            throw FormatError("unsupported attribute version");
        const bool padded = version == 1;
        need(p, nsz);
        name->assign(reinterpret_cast<const char*>(buf_ + p), (size_t)nsz);
        name->resize(std::strlen(name->c_str()));
        p += padded ? pad8(nsz) : nsz;
        const uint64_t dt_off = p;
        p += padded ? pad8(tsz) : tsz;
        const uint64_t ds_off = p;
        p += padded ? pad8(ssz) : ssz;
        const int ds_version = b(ds_off), rank = b(ds_off + 1);
        if ((ds_version != 1 && ds_version != 2) || rank != 0) return false;     // scalars only
        const int cls = b(dt_off) & 0x0F;
        const uint64_t bits = u(dt_off + 1, 3);
        const uint64_t size = u(dt_off + 4, 4);
        const bool big_endian = (bits & 1) != 0;
        a->bytes.clear();
        if (cls == 0 || cls == 1) {
            if (cls == 0 && size != 1 && size != 2 && size != 4 && size != 8) return false;
            if (cls == 1 && size != 4 && size != 8) return false;
            need(p, size);
            a->kind = cls == 0 ? 1 : 2;
            a->size = (int)size;
            a->is_signed = cls == 0 && (bits & 0x08) != 0;
            a->bytes.assign(reinterpret_cast<const char*>(buf_ + p), (size_t)size);
            if (big_endian) std::reverse(a->bytes.begin(), a->bytes.end());
            return true;
        }
        if (cls == 3) {            // fixed-length string, NUL padding stripped like h5py does
            need(p, size);
            a->kind = 0;
            a->bytes.assign(reinterpret_cast<const char*>(buf_ + p), (size_t)size);
            a->bytes.resize(std::strlen(a->bytes.c_str()));
            return true;
        }
        if (cls == 9 && (bits & 0x0F) == 1) {   // variable-length string in the global heap
            need(p, 8 + kO);
            const uint64_t coll = u(p + 4, kO), idx = u(p + 4 + kO, 4);
            a->kind = 0;
            a->bytes = coll ? global_heap_object(coll, idx) : std::string();
            return true;
        }
        return false;
    }

    Attrs attributes(uint64_t header_addr) const {
        Attrs found;
        std::string name;
        Attr a;
        auto take = [&](uint64_t off) {
            try {
                if (parse_attribute(off, &name, &a)) found[name] = a;
                else found.erase(name);
            } catch (const std::exception&) {      // one unreadable attribute is one left behind
            }
        };
        for (const Msg& m : object_header(header_addr)) {
            if (m.type == 0x000C) {
                take(m.off);
            } else if (m.type == 0x0015) {   // dense attribute storage
                const int flags = b(m.off + 1);
                const uint64_t p = m.off + 2 + ((flags & 1) ? 2 : 0);
                const uint64_t heap_addr = u(p, kO), index_addr = u(p + kO, kO);
                if (heap_addr == kUndef || index_addr == kUndef) continue;
                for (uint64_t obj_off : dense_objects(heap_addr, index_addr)) take(obj_off);
            }
        }
        return found;
    }

    // What a one-read copy of read `index` carries beside its Signal (ont_fast5_api's
    // multi_to_single_fast5, the tool the reference runs - realtime.py:183-190 - copies the same):
    // the attributes of the read group, of Raw, and of channel_id / tracking_id / context_tags.
    struct ReadMeta {
        Attrs read, raw, group[3];
        bool has[3] = {false, false, false};
    };
    static const char* meta_group_name(int k) {
        static const char* const names[3] = {"channel_id", "tracking_id", "context_tags"};
        return names[k];
    }
    void read_metadata(int64_t index, ReadMeta* out) {
        const ReadEntry& r = read(index);
        out->raw = attributes(r.group_addr);
        if (r.read_group_addr == 0) return;
        out->read = attributes(r.read_group_addr);
        const std::map<std::string, uint64_t> links = group_links(object_header(r.read_group_addr));
        for (int k = 0; k < 3; ++k) {
            auto it = links.find(meta_group_name(k));
            if (it == links.end()) continue;
            out->group[k] = attributes(it->second);
            out->has[k] = true;
        }
    }

    // The one chunk a read's Signal is stored as, if it is ONE chunk of exactly the read's
    // samples compressed by the deflate filter alone: a one-read copy can carry it as it is.
    bool whole_deflated_chunk(const SignalInfo& s, uint64_t* off, uint64_t* nbytes) const {
        if (s.layout != 2 || s.n <= 0 || s.chunk_elems != s.n || s.filters.size() != 1 ||
            s.filters[0].id != 1)
            return false;
        std::vector<RawPiece> pieces;
        signal_pieces(s, 0, &pieces);
        if (pieces.size() != 1 || pieces[0].kind != kZlib || pieces[0].mask != 0 ||
            pieces[0].count != s.n)
            return false;
        *off = pieces[0].file_off;
        *nbytes = pieces[0].nbytes;
        return true;
    }

  private:
    // ---- the Signal dataset ---------------------------------------------------------------------
    SignalInfo signal_info(uint64_t addr) const {
        const std::vector<Msg> msgs = object_header(addr);
        const Msg* dt = first_of(msgs, 0x0003);
        const Msg* ds = first_of(msgs, 0x0001);
        const Msg* lay = first_of(msgs, 0x0008);
        if (!dt || !ds || !lay) throw FormatError("Signal without datatype/dataspace/layout");
        const int cls = b(dt->off) & 0x0F;
        const uint64_t bits = u(dt->off + 1, 3);
        if (cls != 0 || u(dt->off + 4, 4) != 2 || (bits & 1) || !(bits & 0x08))
            throw FormatError("Signal is not a little-endian int16 dataset");
        const int ds_version = b(ds->off), rank = b(ds->off + 1);
        if ((ds_version != 1 && ds_version != 2) || rank != 1)
            throw FormatError("Signal is not one-dimensional");
        SignalInfo s;
        const uint64_t n = u(ds->off + (ds_version == 1 ? 8 : 4), kL);
        // deflate expands at most 1032-fold, so a signal cannot be much longer than that many
        // times the file (a damaged length would otherwise have the caller allocate terabytes)
        if (n > (1ull << 40) || n / 1100 > len_ + 4096)
            throw FormatError("implausible Signal length");
        s.n = (int64_t)n;

        const int version = b(lay->off);
        if (version == 1 || version == 2) {
            const int ndim = b(lay->off + 1), lcls = b(lay->off + 2);
            uint64_t p = lay->off + 8;
            if (lcls != 0) {
                s.addr = u(p, kO);
                p += kO;
            }
            const uint64_t dims = p;
            p += 4ull * ndim;
            if (lcls == 0) {
                s.layout = 0;
                s.compact_size = u(p, 4);
                s.compact_off = p + 4;
            } else if (lcls == 1) {
                s.layout = 1;
            } else {
                if (ndim != 2) throw FormatError("unexpected chunk rank");
                s.layout = 2;
                s.chunk_elems = (int64_t)u(dims, 4);
            }
        } else if (version == 3) {
            const int lcls = b(lay->off + 1);
            const uint64_t p = lay->off + 2;
            if (lcls == 0) {
                s.layout = 0;
                s.compact_size = u(p, 2);
                s.compact_off = p + 2;
            } else if (lcls == 1) {
                s.layout = 1;
                s.addr = u(p, kO);
            } else if (lcls == 2) {
                if (b(p) != 2) throw FormatError("unexpected chunk rank");
                s.layout = 2;
                s.addr = u(p + 1, kO);
                s.chunk_elems = (int64_t)u(p + 1 + kO, 4);
            } else {
                throw FormatError("unsupported data layout class");
            }
        } else if (version == 4) {
            const int lcls = b(lay->off + 1);
            uint64_t p = lay->off + 2;
            if (lcls == 0) {
                s.layout = 0;
                s.compact_size = u(p, 2);
                s.compact_off = p + 2;
            } else if (lcls == 1) {
                s.layout = 1;
                s.addr = u(p, kO);
            } else if (lcls == 2) {
                const int flags = b(p), ndim = b(p + 1), enc = b(p + 2);
                if (ndim != 2 || enc < 1 || enc > 8) throw FormatError("unexpected chunk rank");
                s.layout = 2;
                s.chunk_elems = (int64_t)u(p + 3, enc);
                s.unfiltered_edge = (flags & 1) != 0;
                p += 3 + 2ull * enc;
                s.index = b(p);
                p += 1;
                if (s.index == 1) {
                    s.single_bytes = (uint64_t)s.chunk_elems * 2;
                    if (flags & 2) {
                        s.single_bytes = u(p, kL);
                        s.single_mask = (uint32_t)u(p + kL, 4);
                        p += kL + 4;
                    }
                } else if (s.index == 2) {
                } else if (s.index == 3) {
                    p += 1;
                } else if (s.index == 4) {
                    p += 5;
                } else {
                    throw FormatError("unsupported chunk index type");
                }
                s.addr = u(p, kO);
            } else {
                throw FormatError("unsupported data layout class");
            }
        } else {
            throw FormatError("unsupported data layout version");
        }

        if (const Msg* fm = first_of(msgs, 0x000B)) {
            const int fversion = b(fm->off), nf = b(fm->off + 1);
            uint64_t p = fm->off + (fversion == 1 ? 8 : 2);
            for (int i = 0; i < nf; ++i) {
                Filter f;
                f.id = (int)u(p, 2);
                p += 2;
                uint64_t name_len = 0;
                if (fversion == 1 || f.id >= 256) {
                    name_len = u(p, 2);
                    p += 2;
                }
                p += 2;   // flags
                const uint64_t ncd = u(p, 2);
                p += 2;
                p += fversion == 1 ? pad8(name_len) : name_len;
                for (uint64_t k = 0; k < ncd; ++k) f.cd.push_back((uint32_t)u(p + 4 * k, 4));
                p += 4 * ncd;
                if (fversion == 1 && (ncd % 2) == 1) p += 4;
                if (f.id != 1 && f.id != 2 && f.id != 3 && s.layout == 2)
                    throw UnsupportedFilter("Signal uses a filter other than deflate/shuffle/fletcher32");
                s.filters.push_back(f);
            }
        }
        return s;
    }

  public:
    // one zlib stream -> its bytes (libdeflate where the system has it, zlib otherwise)
    static void inflate_stream(const uint8_t* src, size_t src_len, size_t hint,
                               std::vector<uint8_t>* out, ChunkCache* cache) {
        inflate_all(src, src_len, hint, out, cache);
    }

  private:
    static void inflate_all(const uint8_t* src, size_t src_len, size_t hint,
                            std::vector<uint8_t>* out, ChunkCache* cache) {
        const LibDeflate& fast = libdeflate();
        if (fast.usable()) {
            if (!cache->fast_inflater) cache->fast_inflater = fast.alloc();
            if (cache->fast_inflater) {
                out->resize(std::max<size_t>(hint, 64));
                size_t produced = 0;
                if (fast.zlib_decompress(cache->fast_inflater, src, src_len, out->data(),
                                         out->size(), &produced) == 0) {
                    out->resize(produced);
                    return;
                }
            }
        }
        z_stream& zs = cache->zs;
        if (cache->zs_ready) {
            if (inflateReset(&zs) != Z_OK) throw FormatError("zlib reset failed");
        } else {
            std::memset(&zs, 0, sizeof(zs));
            if (inflateInit(&zs) != Z_OK) throw FormatError("zlib init failed");
            cache->zs_ready = true;
        }
        out->resize(std::max<size_t>(hint, 64));
        zs.next_in = const_cast<Bytef*>(src);
        zs.avail_in = (uInt)src_len;
        size_t produced = 0;
        while (true) {
            zs.next_out = out->data() + produced;
            zs.avail_out = (uInt)(out->size() - produced);
            const int rc = inflate(&zs, Z_NO_FLUSH);
            produced = out->size() - zs.avail_out;
            if (rc == Z_STREAM_END) break;
            if (rc != Z_OK || (zs.avail_in == 0 && zs.avail_out != 0))
                throw FormatError("corrupt deflate stream");
            if (zs.avail_out == 0) {
                if (out->size() > (1u << 30))
                    throw FormatError("chunk inflates to an implausible size");
                out->resize(out->size() * 2);
            }
        }
        out->resize(produced);
    }

    // one stored chunk -> its elements (filters undone in reverse order, short chunks zero-extended)
    void decode_chunk(const SignalInfo& s, uint64_t addr, uint64_t nbytes, uint32_t mask,
                      ChunkCache* cache) const {
        std::vector<uint8_t>* raw = &cache->data;
        std::vector<uint8_t>& tmp = cache->scratch;
        const uint64_t start = file_off(addr);
        need(start, nbytes);
        if (mapped_ && fd_ >= 0) {
            // large (multi-read) files: the payload comes through pread, not through the
            // mapping - page faults of many threads queue on the address-space lock
            raw->resize((size_t)nbytes);
            uint64_t got = 0;
            while (got < nbytes) {
                const ssize_t k = ::pread(fd_, raw->data() + got, (size_t)(nbytes - got),
                                          (off_t)(start + got));
                if (k <= 0) throw FormatError("cannot read chunk");
                got += (uint64_t)k;
            }
        } else {
            raw->assign(buf_ + start, buf_ + start + nbytes);
        }
        for (int i = (int)s.filters.size() - 1; i >= 0; --i) {
            if (mask & (1u << i)) continue;
            const Filter& f = s.filters[(size_t)i];
            if (f.id == 1) {
                inflate_all(raw->data(), raw->size(), (size_t)s.chunk_elems * 2 + 8, &tmp, cache);
                raw->swap(tmp);
            } else if (f.id == 2) {
                const size_t esize = f.cd.empty() ? 2 : f.cd[0];
                if (esize == 0) throw FormatError("bad shuffle parameter");
                const size_t n = raw->size() / esize;
                tmp.assign(raw->begin(), raw->end());
                for (size_t e = 0; e < esize; ++e)
                    for (size_t k = 0; k < n; ++k) tmp[k * esize + e] = (*raw)[e * n + k];
                raw->swap(tmp);
            } else if (f.id == 3) {
                if (raw->size() < 4) throw FormatError("chunk shorter than its checksum");
                raw->resize(raw->size() - 4);
            } else {
                throw FormatError("unsupported filter");
            }
        }
        // some writers (MinKNOW) store a short final chunk; libhdf5 zero-extends it
        if (raw->size() < (size_t)s.chunk_elems * 2) raw->resize((size_t)s.chunk_elems * 2, 0);
    }

    // the part of [first, first + count) that the chunk starting at sample `lo` holds
    void copy_from_chunk(const SignalInfo& s, int64_t lo, uint64_t addr, uint64_t nbytes,
                         uint32_t mask, int64_t first, int64_t count, int16_t* out,
                         ChunkCache* cache) const {
        const int64_t hi = std::min<int64_t>(lo + s.chunk_elems, s.n);
        const int64_t a = std::max(lo, first), z = std::min(hi, first + count);
        if (a >= z) return;
        // the last chunk inflated stays around: a read stored as ONE chunk (common) is asked
        // for twice, once per end, and deflate cannot be entered in the middle
        if (cache->owner != this || cache->addr != addr || cache->bytes != nbytes) {
            cache->addr = ~0ull;
            decode_chunk(s, addr, nbytes, mask, cache);
            cache->owner = this;
            cache->addr = addr;
            cache->bytes = nbytes;
        }
        std::memcpy(out + (a - first), cache->data.data() + (size_t)(a - lo) * 2, (size_t)(z - a) * 2);
    }

    // ---- chunk indexes of layout version 4 (hdf5_lite._fixed_array / _extensible_array) ----------
    // one record of a fixed / extensible array; false: the chunk was never written
    bool index_record(uint64_t p, bool filtered, int elmt_size, uint64_t chunk_bytes,
                      uint64_t* addr, uint64_t* nbytes, uint32_t* mask) const {
        *addr = u(p, kO);
        if (*addr == kUndef) return false;
        if (!filtered) {
            *nbytes = chunk_bytes;
            *mask = 0;
            return true;
        }
        const int size_len = elmt_size - kO - 4;
        if (size_len < 1 || size_len > 8) throw FormatError("bad chunk record size");
        *nbytes = u(p + kO, size_len);
        *mask = (uint32_t)u(p + kO + size_len, 4);
        return true;
    }

    static int log2_floor(uint64_t v) {
        int r = -1;
        while (v) {
            v >>= 1;
            ++r;
        }
        return r;
    }

    bool bit_set(uint64_t bitmap, uint64_t bit) const { return (b(bitmap + bit / 8) & (0x80u >> (bit % 8))) != 0; }

    bool indexed_chunk(const SignalInfo& s, uint64_t k, uint64_t* addr, uint64_t* nbytes,
                       uint32_t* mask) const {
        const uint64_t chunk_bytes = (uint64_t)s.chunk_elems * 2;
        if (s.index == 1) {
            if (k != 0) return false;
            *addr = s.addr;
            *nbytes = s.single_bytes;
            *mask = s.single_mask;
            return true;
        }
        if (s.index == 2) {
            *addr = s.addr + k * chunk_bytes;
            *nbytes = chunk_bytes;
            *mask = 0;
            return true;
        }
        const uint64_t p = file_off(s.addr);
        if (s.index == 3) {
            if (!sig(p, "FAHD") || b(p + 4) != 0) throw FormatError("bad fixed array header");
            const bool filtered = b(p + 5) == 1;
            const int elmt_size = b(p + 6), page_bits = b(p + 7);
            const uint64_t n = u(p + 8, kL);
            if (k >= n || page_bits > 40) return false;
            const uint64_t block = file_off(u(p + 8 + kL, kO));
            if (!sig(block, "FADB")) throw FormatError("bad fixed array data block");
            uint64_t q = block + 6 + kO;
            const uint64_t page = 1ull << page_bits;
            if (n > page) {            // paged: bitmap, checksum, then checksummed pages
                const uint64_t n_pages = (n + page - 1) / page;
                const uint64_t pg = k / page, within = k % page;
                if (!bit_set(q, pg)) return false;
                q += (n_pages + 7) / 8 + 4;
                return index_record(q + pg * (page * elmt_size + 4) + within * elmt_size, filtered,
                                    elmt_size, chunk_bytes, addr, nbytes, mask);
            }
            return index_record(q + k * elmt_size, filtered, elmt_size, chunk_bytes, addr, nbytes, mask);
        }
        // extensible array: the first records in the index block, then data blocks of doubling size,
        // the early ones addressed from the index block, the later ones through super blocks
        if (!sig(p, "EAHD") || b(p + 4) != 0) throw FormatError("bad extensible array header");
        const bool filtered = b(p + 5) == 1;
        const int elmt_size = b(p + 6), max_bits = b(p + 7), idx_elmts = b(p + 8),
                  dblk_min = b(p + 9), sblk_min_ptrs = b(p + 10), page_bits = b(p + 11);
        if (dblk_min < 1 || sblk_min_ptrs < 2 || max_bits > 64 || page_bits > 40 ||
            (dblk_min & (dblk_min - 1)) || (sblk_min_ptrs & (sblk_min_ptrs - 1)))
            throw FormatError("bad extensible array parameters");
        const uint64_t index_block = u(p + 12 + 6 * kL, kO);
        if (index_block == kUndef) return false;
        const uint64_t q = file_off(index_block);
        if (!sig(q, "EAIB")) throw FormatError("bad extensible array index block");
        const uint64_t elements = q + 6 + kO;
        if (k < (uint64_t)idx_elmts)
            return index_record(elements + k * elmt_size, filtered, elmt_size, chunk_bytes, addr,
                                nbytes, mask);
        const int off_size = (max_bits + 7) / 8;
        const int n_sblks = 1 + max_bits - log2_floor((uint64_t)dblk_min);
        const int iblock_sblks = 2 * log2_floor((uint64_t)sblk_min_ptrs);
        const uint64_t n_dblk_addrs = 2ull * (sblk_min_ptrs - 1);
        const uint64_t dblk_addrs = elements + (uint64_t)idx_elmts * elmt_size;
        const uint64_t sblk_addrs = dblk_addrs + n_dblk_addrs * kO;
        uint64_t rel = k - idx_elmts;
        const int s_idx = log2_floor(rel / dblk_min + 1);
        if (s_idx >= n_sblks || s_idx > 62) return false;
        // super block u: 2^(u/2) data blocks of 2^((u+1)/2) * dblk_min records
        uint64_t first_elmt = 0, first_dblk = 0;
        for (int v = 0; v < s_idx; ++v) {
            first_elmt += (1ull << (v / 2)) * ((1ull << ((v + 1) / 2)) * dblk_min);
            first_dblk += 1ull << (v / 2);
        }
        const uint64_t n_dblks = 1ull << (s_idx / 2);
        const uint64_t dblk_elmts = (1ull << ((s_idx + 1) / 2)) * dblk_min;
        rel -= first_elmt;
        const uint64_t d_idx = rel / dblk_elmts, within = rel % dblk_elmts;
        const uint64_t page = 1ull << page_bits;
        uint64_t block_addr;
        if (s_idx < iblock_sblks) {
            block_addr = u(dblk_addrs + (first_dblk + d_idx) * kO, kO);
        } else {
            const uint64_t super_addr = u(sblk_addrs + (uint64_t)(s_idx - iblock_sblks) * kO, kO);
            if (super_addr == kUndef) return false;
            const uint64_t sb = file_off(super_addr);
            if (!sig(sb, "EASB")) throw FormatError("bad extensible array super block");
            uint64_t t = sb + 6 + kO + off_size;
            if (dblk_elmts > page) {
                // "page initialised" bits: a whole number of bytes per data block is reserved,
                // but the bits are used as one run, `pages` per data block
                const uint64_t pages = dblk_elmts / page;
                if (!bit_set(t, d_idx * pages + within / page)) return false;
                t += n_dblks * ((pages + 7) / 8);
            }
            block_addr = u(t + d_idx * kO, kO);
        }
        if (block_addr == kUndef) return false;
        const uint64_t blk = file_off(block_addr);
        if (!sig(blk, "EADB")) throw FormatError("bad extensible array data block");
        uint64_t e = blk + 6 + kO + off_size;
        uint64_t at = within;
        if (dblk_elmts > page) {       // prefix checksum, then checksummed pages
            e += 4 + (within / page) * (page * elmt_size + 4);
            at = within % page;
        }
        return index_record(e + at * elmt_size, filtered, elmt_size, chunk_bytes, addr, nbytes, mask);
    }

    // chunk index: version-1 B-tree of raw-data chunks, rank 1 (hdf5_lite._walk_chunk_btree)
    void walk_chunks(const SignalInfo& s, uint64_t addr, int64_t first, int64_t count, int16_t* out,
                     int depth, ChunkCache* cache) const {
        if (depth > kMaxDepth) throw FormatError("chunk B-tree too deep");
        const uint64_t p = file_off(addr);
        if (!sig(p, "TREE")) throw FormatError("bad chunk B-tree signature");
        if (b(p + 4) != 1) throw FormatError("expected a raw-data chunk B-tree");
        const int level = b(p + 5);
        const uint64_t used = u(p + 6, 2);
        const uint64_t key_size = 8 + 8 * 2;
        const uint64_t q = p + 8 + 2 * kO;
        for (uint64_t i = 0; i < used; ++i) {
            const uint64_t k = q + i * (key_size + kO);
            const uint64_t nbytes = u(k, 4);
            const uint32_t mask = (uint32_t)u(k + 4, 4);
            const uint64_t offset = u(k + 8, 8);
            const uint64_t child = u(k + key_size, kO);
            if (level > 0) {
                // keys are the first chunk offsets of the children: skip subtrees wholly outside
                if (i + 1 < used) {
                    const uint64_t next = u(q + (i + 1) * (key_size + kO) + 8, 8);
                    if ((int64_t)next <= first) continue;
                }
                if ((int64_t)offset >= first + count) break;
                walk_chunks(s, child, first, count, out, depth + 1, cache);
                continue;
            }
            if (offset >= (uint64_t)s.n) continue;
            copy_from_chunk(s, (int64_t)offset, child, nbytes, mask, first, count, out, cache);
        }
    }

    // the same walk, reporting every chunk instead of copying from it
    template <class Found>
    void collect_chunks(const SignalInfo& s, uint64_t addr, int depth, const Found& found) const {
        if (depth > kMaxDepth) throw FormatError("chunk B-tree too deep");
        const uint64_t p = file_off(addr);
        if (!sig(p, "TREE")) throw FormatError("bad chunk B-tree signature");
        if (b(p + 4) != 1) throw FormatError("expected a raw-data chunk B-tree");
        const int level = b(p + 5);
        const uint64_t used = u(p + 6, 2);
        const uint64_t key_size = 8 + 8 * 2;
        const uint64_t q = p + 8 + 2 * kO;
        for (uint64_t i = 0; i < used; ++i) {
            const uint64_t k = q + i * (key_size + kO);
            const uint64_t nbytes = u(k, 4);
            const uint32_t mask = (uint32_t)u(k + 4, 4);
            const uint64_t offset = u(k + 8, 8);
            const uint64_t child = u(k + key_size, kO);
            if (level > 0) {
                collect_chunks(s, child, depth + 1, found);
                continue;
            }
            if (offset >= (uint64_t)s.n || offset % (uint64_t)s.chunk_elems) continue;
            found((int64_t)(offset / (uint64_t)s.chunk_elems), child, nbytes, mask);
        }
    }

    // ---- which reads the file holds (load_fast5s.py:29-43) ---------------------------------------
    void find_reads() {
        const std::map<std::string, uint64_t> root = group_links(object_header(root_addr_));
        auto raw = root.find("Raw");
        if (raw != root.end()) {   // older format: exactly one read under /Raw/Reads
            const std::map<std::string, uint64_t> raw_links = group_links(object_header(raw->second));
            auto reads = raw_links.find("Reads");
            if (reads == raw_links.end()) throw std::out_of_range("no /Raw/Reads");
            const std::map<std::string, uint64_t> children = group_links(object_header(reads->second));
            if (children.empty()) throw std::out_of_range("empty /Raw/Reads");
            ReadEntry e;
            e.group_addr = children.begin()->second;
            reads_.push_back(e);
            layout_ = F5_LAYOUT_SINGLE_OLD;
            return;
        }
        for (const auto& kv : root) {
            if (kv.first.compare(0, 5, "read_") != 0) continue;
            const std::map<std::string, uint64_t> links = group_links(object_header(kv.second));
            auto r = links.find("Raw");
            if (r == links.end()) throw std::out_of_range("read group without Raw");
            ReadEntry e;
            e.group_addr = r->second;
            e.read_group_addr = kv.second;
            reads_.push_back(e);
        }
        layout_ = reads_.empty() ? F5_LAYOUT_NONE
                                 : (reads_.size() == 1 ? F5_LAYOUT_SINGLE_NEW : F5_LAYOUT_MULTI);
    }
};

// Runs fn and maps what it throws to the ABI's status codes.
template <class Fn>
int guarded(Fn&& fn) {
    try {
        fn();
        return F5_OK;
    } catch (const UnsupportedFilter&) {
        return F5_ERR_FILTER;
    } catch (const ExistsError&) {
        return F5_ERR_EXISTS;
    } catch (const FormatError&) {
        return F5_ERR_FORMAT;
    } catch (const std::out_of_range&) {
        return F5_ERR_NO_READ;
    } catch (const std::bad_alloc&) {
        return F5_ERR_FORMAT;
    } catch (const std::exception&) {
        return F5_ERR_OPEN;
    }
}

void copy_read_id(const std::string& id, char* dst) {
    std::memset(dst, 0, F5_READ_ID_MAX);
    std::memcpy(dst, id.data(), std::min<size_t>(id.size(), F5_READ_ID_MAX - 1));
}

// ---------------------------------------------------------------------------------------------
// Writing one-read fast5 files: the layout of deepbinner_amd/hdf5_write.py (superblock version 0,
// version-1 object headers, groups as symbol tables, the Signal as one deflate-compressed chunk
// behind a version-1 B-tree, scalar attributes), byte for byte - that module is pinned to the
// real HDF5 library and to the reference's loader (tests/test_hdf5_write.py), and
// tests/test_fast5_writer.py holds this one to it.  Why twice: `deepbinner realtime` ends with
// every read of a multi-read container as a one-read file in its barcode's directory
// (reference realtime.py:111-150 after multi_to_single_fast5, :183-190); built in Python that is
// ~350 us per read under the interpreter lock plus a deflate of the signal - 2.5 k reads/s
// behind a GPU that classifies 170 k.  Here a read costs its attributes, one pread of its chunk
// AS STORED (nothing is inflated, nothing deflated again) and one write.
// ---------------------------------------------------------------------------------------------
namespace h5w {

constexpr uint64_t kUndefAddr = 0xFFFFFFFFFFFFFFFFull;
constexpr int kGroupLeafK = 4, kGroupInternalK = 16, kChunkK = 32;      // superblock v0 defaults

struct Bytes {
    std::string s;
    Bytes& u8(uint64_t v) { return put(v, 1); }
    Bytes& u16(uint64_t v) { return put(v, 2); }
    Bytes& u32(uint64_t v) { return put(v, 4); }
    Bytes& u64(uint64_t v) { return put(v, 8); }
    Bytes& zeros(size_t n) {
        s.append(n, '\0');
        return *this;
    }
    Bytes& raw(const std::string& b) {
        s += b;
        return *this;
    }
    Bytes& raw(const char* b, size_t n) {
        s.append(b, n);
        return *this;
    }
    Bytes& pad8() { return zeros((size_t)((8 - s.size() % 8) % 8)); }
    Bytes& put(uint64_t v, int n) {
        for (int i = 0; i < n; ++i) s.push_back((char)((v >> (8 * i)) & 0xFF));
        return *this;
    }
};
std::string padded8(const std::string& b) { return Bytes{b}.pad8().s; }

// the file as one growing byte string; every block starts 8-byte aligned
struct Image {
    std::string buf;
    uint64_t reserve(size_t size) {
        buf.append((8 - buf.size() % 8) % 8, '\0');
        const uint64_t addr = buf.size();
        buf.append(size, '\0');
        return addr;
    }
    uint64_t add(const std::string& data) {
        const uint64_t addr = reserve(data.size());
        std::memcpy(&buf[(size_t)addr], data.data(), data.size());
        return addr;
    }
};

std::string message(int type, const std::string& body, int flags = 0) {
    const std::string b = padded8(body);
    return Bytes().u16((uint64_t)type).u16(b.size()).u8((uint64_t)flags).zeros(3).raw(b).s;
}
std::string object_header(const std::vector<std::string>& messages) {
    std::string data;
    for (const std::string& m : messages) data += m;
    return Bytes().u8(1).u8(0).u16(messages.size()).u32(1).u32(data.size()).zeros(4).raw(data).s;
}
std::string string_datatype(size_t size) { return Bytes().u8(0x13).zeros(3).u32(size).s; }
std::string fixed_datatype(int size, bool is_signed) {
    return Bytes().u8(0x10).u8(is_signed ? 0x08 : 0).zeros(2).u32((uint64_t)size).u16(0)
        .u16((uint64_t)(8 * size)).s;
}
std::string float_datatype(int size) {
    if (size == 4)
        return Bytes().u8(0x11).u8(0x20).u8(0x1F).u8(0).u32(4).u16(0).u16(32).u8(23).u8(8).u8(0)
            .u8(23).u32(127).s;
    return Bytes().u8(0x11).u8(0x20).u8(0x3F).u8(0).u32(8).u16(0).u16(64).u8(52).u8(11).u8(0)
        .u8(52).u32(1023).s;
}
std::string scalar_dataspace() { return Bytes().u8(1).u8(0).u8(0).zeros(5).s; }
std::string simple_dataspace(uint64_t n) { return Bytes().u8(1).u8(1).u8(0).zeros(5).u64(n).s; }
std::string attribute(const std::string& name, const std::string& datatype,
                      const std::string& value) {
    const std::string name_z = name + std::string(1, '\0');
    const std::string space = scalar_dataspace();
    return message(0x000C, Bytes().u8(1).u8(0).u16(name_z.size()).u16(datatype.size())
                               .u16(space.size()).raw(padded8(name_z)).raw(padded8(datatype))
                               .raw(padded8(space)).raw(value).s);
}
std::string string_attribute(const std::string& name, const std::string& text) {
    const std::string raw = text + std::string(1, '\0');
    return attribute(name, string_datatype(raw.size()), raw);
}
std::string int_attribute(const std::string& name, uint64_t value, int size, bool is_signed) {
    return attribute(name, fixed_datatype(size, is_signed), Bytes().put(value, size).s);
}
std::string attribute_of(const std::string& name, const Fast5::Attr& a) {
    if (a.kind == 0) return string_attribute(name, a.bytes);
    if (a.kind == 1) return attribute(name, fixed_datatype(a.size, a.is_signed), a.bytes);
    return attribute(name, float_datatype(a.size), a.bytes);
}
std::vector<std::string> attributes_of(const Fast5::Attrs& values, const char* skip = nullptr) {
    std::vector<std::string> out;            // (a std::map walks its names in sorted order)
    for (const auto& kv : values)
        if (!skip || kv.first != skip) out.push_back(attribute_of(kv.first, kv.second));
    return out;
}

struct Node {
    uint64_t header = 0, btree = kUndefAddr, heap = kUndefAddr;
    bool is_group = true;
};

Node group(Image& image, const std::map<std::string, Node>& children,
           const std::vector<std::string>& attributes) {
    // local heap data: offset 0 is the empty string, then the names, each 8-byte aligned
    // (a std::map walks the names in byte order, which is what the symbol table wants)
    Bytes heap_data;
    heap_data.zeros(8);
    std::vector<uint64_t> offsets;
    for (const auto& kv : children) {
        offsets.push_back(heap_data.s.size());
        heap_data.raw(padded8(kv.first + std::string(1, '\0')));
    }
    const uint64_t heap_data_addr = image.add(heap_data.s);
    const uint64_t heap = image.add(
        Bytes().raw("HEAP", 4).u8(0).zeros(3).u64(heap_data.s.size()).u64(1).u64(heap_data_addr).s);
    Bytes snod;
    snod.raw("SNOD", 4).u8(1).u8(0).u16(children.size());
    size_t k = 0;
    for (const auto& kv : children) {
        const Node& c = kv.second;
        snod.u64(offsets[k++]).u64(c.header);
        if (c.is_group)
            snod.u32(1).u32(0).u64(c.btree).u64(c.heap);
        else
            snod.u32(0).u32(0).zeros(16);
    }
    snod.zeros(40 * (2 * kGroupLeafK - children.size()));
    const uint64_t snod_addr = image.add(snod.s);
    // B-tree v1, group node (type 0), leaf level, one child: keys are heap offsets of names
    Bytes node;
    node.raw("TREE", 4).u8(0).u8(0).u16(1).u64(kUndefAddr).u64(kUndefAddr);
    node.u64(0).u64(snod_addr).u64(offsets.back());
    node.zeros(24 + 16 * kGroupInternalK * 2 + 8 - node.s.size());
    Node g;
    g.btree = image.add(node.s);
    g.heap = heap;
    std::vector<std::string> messages = {message(0x0011, Bytes().u64(g.btree).u64(heap).s)};
    messages.insert(messages.end(), attributes.begin(), attributes.end());
    g.header = image.add(object_header(messages));
    return g;
}

// a group without links that only carries attributes (channel_id, tracking_id, ...)
Node attribute_group(Image& image, const std::vector<std::string>& attributes) {
    const uint64_t heap_data_addr = image.add(std::string(8, '\0'));
    Node g;
    g.heap = image.add(Bytes().raw("HEAP", 4).u8(0).zeros(3).u64(8).u64(1).u64(heap_data_addr).s);
    Bytes node;
    node.raw("TREE", 4).u8(0).u8(0).u16(0).u64(kUndefAddr).u64(kUndefAddr);
    node.zeros(24 + 16 * kGroupInternalK * 2 + 8 - node.s.size());
    g.btree = image.add(node.s);
    std::vector<std::string> messages = {message(0x0011, Bytes().u64(g.btree).u64(g.heap).s)};
    messages.insert(messages.end(), attributes.begin(), attributes.end());
    g.header = image.add(object_header(messages));
    return g;
}

// int16[n] as one deflate-compressed chunk (`packed`: its zlib stream) behind a chunk B-tree
Node dataset_int16(Image& image, uint64_t n, const std::string& packed) {
    std::vector<std::string> messages = {
        message(0x0001, simple_dataspace(n)),
        message(0x0003, fixed_datatype(2, true), 1),                   // constant message
        message(0x0005, Bytes().u8(2).u8(2).u8(0).u8(0).s)};           // fill value: never written
    if (n > 0) {
        const uint64_t chunk = image.add(packed);
        const size_t key_size = 8 + 8 * 2;
        Bytes node;
        node.raw("TREE", 4).u8(1).u8(0).u16(1).u64(kUndefAddr).u64(kUndefAddr);
        node.u32(packed.size()).u32(0).u64(0).u64(0).u64(chunk);
        node.u32(0).u32(0).u64(n).u64(0);
        node.zeros(24 + 2 * kChunkK * 8 + (2 * kChunkK + 1) * key_size - node.s.size());
        const uint64_t btree = image.add(node.s);
        messages.push_back(message(
            0x000B, Bytes().u8(1).u8(1).zeros(6).u16(1).u16(0).u16(0).u16(1).u32(1).u32(0).s));
        messages.push_back(message(0x0008, Bytes().u8(3).u8(2).u8(2).u64(btree).u32(n).u32(2).s));
    } else {
        messages.push_back(message(0x0008, Bytes().u8(3).u8(1).u64(kUndefAddr).u64(0).s));
    }
    Node d;
    d.is_group = false;
    d.header = image.add(object_header(messages));
    return d;
}

// The bytes of a one-read fast5 file: /read_<id>/{Raw/Signal, channel_id, tracking_id,
// context_tags} with the attributes of `meta` (hdf5_write.single_read_fast5_bytes).
std::string single_read_file(const std::string& read_id, uint64_t n_samples,
                             const std::string& packed, const Fast5::ReadMeta& meta) {
    Image image;
    const uint64_t superblock = image.reserve(96);
    const Node signal = dataset_int16(image, n_samples, packed);
    std::vector<std::string> raw_attributes = {string_attribute("read_id", read_id)};
    if (meta.raw.find("duration") == meta.raw.end())
        raw_attributes.push_back(int_attribute("duration", n_samples, 4, false));
    for (const std::string& a : attributes_of(meta.raw, "read_id")) raw_attributes.push_back(a);
    std::map<std::string, Node> children;
    children["Raw"] = group(image, {{"Signal", signal}}, raw_attributes);
    for (int k = 0; k < 3; ++k)
        if (meta.has[k])
            children[Fast5::meta_group_name(k)] =
                attribute_group(image, attributes_of(meta.group[k]));
    const Node read = group(image, children, attributes_of(meta.read));
    const Node root = group(image, {{"read_" + read_id, read}},
                            {string_attribute("file_version", "2.0")});
    image.buf.append((8 - image.buf.size() % 8) % 8, '\0');
    const uint64_t end = image.buf.size();
    const std::string head =
        Bytes().raw("\x89HDF\r\n\x1a\n", 8).zeros(5).u8(8).u8(8).u8(0).u16(kGroupLeafK)
            .u16(kGroupInternalK).u32(0).u64(0).u64(kUndefAddr).u64(end).u64(kUndefAddr)
            .u64(0).u64(root.header).u32(1).u32(0).u64(root.btree).u64(root.heap).s;
    std::memcpy(&image.buf[(size_t)superblock], head.data(), head.size());
    return std::move(image.buf);
}

}  // namespace h5w

}  // namespace

// Worker threads that outlive the call: a batch of 256 one-read files is 3-4 ms of work, and
// starting and joining 32-64 threads twice per batch was a third of it.  One job at a time; a
// caller that finds the pool busy (several threads in f5_load_batch at once) starts threads of its
// own as before.  The calling thread always works too (slot 0).
class WorkerPool {
  public:
    using Job = std::function<void(int64_t item, int slot)>;

    ~WorkerPool() {
        if (getpid() != owner_) {      // a forked child: the threads exist in the parent only
            for (std::thread& t : workers_) t.detach();
            return;
        }
        {
            std::lock_guard<std::mutex> g(m_);
            stop_ = true;
        }
        wake_.notify_all();
        for (std::thread& t : workers_) t.join();
    }

    // items 0 .. count-1, each exactly once, on up to `threads` threads; slot < threads
    void run(int threads, int64_t count, const Job& job) {
        threads = (int)std::max<int64_t>(1, std::min<int64_t>(threads, count));
        if (threads == 1) {
            for (int64_t i = 0; i < count; ++i) job(i, 0);
            return;
        }
        std::unique_lock<std::mutex> owner(busy_, std::try_to_lock);
        if (!owner.owns_lock() || getpid() != owner_) {
            run_on_new_threads(threads, count, job);
            return;
        }
        {
            std::lock_guard<std::mutex> g(m_);
            while ((int)workers_.size() < threads - 1) {
                const int slot = (int)workers_.size() + 1;
                workers_.emplace_back([this, slot] { worker(slot); });
            }
            job_ = &job;
            count_ = count;
            next_.store(0);
            helpers_ = threads - 1;
            unfinished_ = threads - 1;
            ++generation_;
        }
        wake_.notify_all();
        drain(job, 0);
        std::unique_lock<std::mutex> g(m_);
        done_.wait(g, [this] { return unfinished_ == 0; });
        job_ = nullptr;
    }

  private:
    void drain(const Job& job, int slot) {
        for (int64_t i = next_.fetch_add(1); i < count_; i = next_.fetch_add(1)) {
            try {
                job(i, slot);
            } catch (...) {    // jobs report through their own status fields; never unwind a worker
            }
        }
    }
    void worker(int slot) {
        uint64_t seen = 0;
        for (;;) {
            const Job* job = nullptr;
            {
                std::unique_lock<std::mutex> g(m_);
                wake_.wait(g, [&] { return stop_ || (generation_ != seen && slot <= helpers_); });
                if (stop_) return;
                seen = generation_;
                job = job_;
            }
            drain(*job, slot);
            {
                std::lock_guard<std::mutex> g(m_);
                --unfinished_;
            }
            done_.notify_one();
        }
    }
    static void run_on_new_threads(int threads, int64_t count, const Job& job) {
        std::atomic<int64_t> next(0);
        auto body = [&](int slot) {
            for (int64_t i = next.fetch_add(1); i < count; i = next.fetch_add(1)) job(i, slot);
        };
        std::vector<std::thread> extra;
        for (int t = 1; t < threads; ++t) extra.emplace_back(body, t);
        body(0);
        for (std::thread& t : extra) t.join();
    }

    const pid_t owner_ = getpid();
    std::mutex busy_, m_;
    std::condition_variable wake_, done_;
    std::vector<std::thread> workers_;
    const Job* job_ = nullptr;
    int64_t count_ = 0;
    std::atomic<int64_t> next_{0};
    int helpers_ = 0, unfinished_ = 0;
    uint64_t generation_ = 0;
    bool stop_ = false;
};

// How many hardware threads this process may really keep busy: the online CPUs, cut down to the
// scheduler's affinity mask and to the cgroup's CPU quota (cpu.max of cgroup v2, cfs_quota_us of
// v1).  A container that shows 256 CPUs with a quota of 16 runs a 128-thread team SLOWER than a
// 16-thread one - every thread beyond the quota only adds throttling stalls
// (profiles/r03_cpu_capacity.txt).
int usable_cpus() {
    static const int n = [] {
        int cpus = (int)std::thread::hardware_concurrency();
        if (cpus < 1) cpus = 1;
        cpu_set_t mask;
        if (sched_getaffinity(0, sizeof(mask), &mask) == 0) {
            const int allowed = CPU_COUNT(&mask);
            if (allowed > 0) cpus = std::min(cpus, allowed);
        }
        long long quota = -1, period = -1;
        if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char first[32] = {0};
            if (std::fscanf(f, "%31s %lld", first, &period) == 2 && std::strcmp(first, "max") != 0)
                quota = std::atoll(first);
            std::fclose(f);
        } else {
            if (FILE* q = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
                if (std::fscanf(q, "%lld", &quota) != 1) quota = -1;
                std::fclose(q);
            }
            if (FILE* p = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
                if (std::fscanf(p, "%lld", &period) != 1) period = -1;
                std::fclose(p);
            }
        }
        if (quota > 0 && period > 0)
            cpus = std::min<long long>(cpus, std::max<long long>(1, (quota + period - 1) / period));
        return cpus;
    }();
    return n;
}

// n_threads as the ABI takes it: <= 0 = one per usable hardware thread, at most 64; an explicit
// count is honoured up to 256
int thread_count(int n_threads) {
    if (n_threads > 0) return std::min(n_threads, 256);
    return std::max(1, std::min(usable_cpus(), 64));
}

WorkerPool& worker_pool() {
    static WorkerPool pool;
    return pool;
}

struct f5_file {
    Fast5 impl;
    explicit f5_file(const char* path) : impl(path) {}
};

// Packed samples of a batch.  Deliberately NOT value-initialised (a std::vector would zero 100+ MB
// on the calling thread before the workers start), and RECYCLED: a 4,000-read container's scanned
// ends are 106 MB, and a fresh allocation of that size is an mmap whose 26,000 pages are then
// faulted in by the worker threads one by one and unmapped again when the batch is freed.  Freed buffers wait in a pool (bounded: DEEPBINNER_FAST5_POOL_MB, default 2048)
// for the next batch that fits.  The memory comes from malloc or from an allocator the caller
// installs (f5_set_sample_allocator) - pinned host memory, so that the GPU's DMA engine reads the
// batch where the loader threads wrote it.
struct SampleAllocator {
    f5_alloc_fn alloc = nullptr;
    f5_free_fn release = nullptr;
    void* user = nullptr;
    uint64_t generation = 0;
};

class SamplePool {
  public:
    struct Block {
        void* ptr = nullptr;
        size_t bytes = 0;
        SampleAllocator from;
    };
    ~SamplePool() { flush(); }

    void set_allocator(f5_alloc_fn alloc, f5_free_fn release, void* user) {
        std::vector<Block> old;
        {
            std::lock_guard<std::mutex> g(m_);
            allocator_.alloc = alloc;
            allocator_.release = release;
            allocator_.user = user;
            ++allocator_.generation;
            old.swap(idle_);
            idle_bytes_ = 0;
        }
        for (Block& b : old) free_block(b);
    }

    Block take(size_t bytes) {
        if (bytes == 0) return Block();
        SampleAllocator from;
        {
            std::lock_guard<std::mutex> g(m_);
            // best fit among the idle blocks; one that is far too large stays for a larger batch
            size_t best = idle_.size();
            for (size_t i = 0; i < idle_.size(); ++i)
                if (idle_[i].bytes >= bytes && idle_[i].bytes / 4 <= bytes + (1u << 20) &&
                    (best == idle_.size() || idle_[i].bytes < idle_[best].bytes))
                    best = i;
            if (best != idle_.size()) {
                Block b = idle_[best];
                idle_.erase(idle_.begin() + (std::ptrdiff_t)best);     // (the front stays the oldest)
                idle_bytes_ -= b.bytes;
                return b;
            }
            from = allocator_;
        }
        // more than asked for: the next container is about, not exactly, as large - and a fresh
        // buffer costs its page faults (3-4 times the copy that fills it)
        Block b;
        b.bytes = (bytes + bytes / 4 + (1u << 20)) & ~(size_t)((1u << 20) - 1);
        b.from = from;
        b.ptr = from.alloc ? from.alloc(b.bytes, from.user) : std::malloc(b.bytes);
        if (!b.ptr) throw std::bad_alloc();
        return b;
    }

    void give(Block b) {
        if (!b.ptr) return;
        // The block just used is the size the caller needs NOW: it stays, and the blocks that
        // have waited longest go if the pool is over its limit (a pool full of another
        // workload's sizes would otherwise never serve a request nor take a block back - every
        // batch a fresh allocation and a free, and freeing pinned memory waits for the GPU).
        std::vector<Block> evicted;
        {
            std::lock_guard<std::mutex> g(m_);
            if (b.from.generation == allocator_.generation && b.bytes <= limit()) {
                idle_.push_back(b);
                idle_bytes_ += b.bytes;
                b = Block();
                size_t n = 0;
                while (idle_bytes_ > limit() && n + 1 < idle_.size()) {
                    idle_bytes_ -= idle_[n].bytes;
                    evicted.push_back(idle_[n++]);
                }
                idle_.erase(idle_.begin(), idle_.begin() + (std::ptrdiff_t)n);
            }
        }
        for (Block& old : evicted) free_block(old);
        free_block(b);
    }

    void flush() {
        std::vector<Block> old;
        {
            std::lock_guard<std::mutex> g(m_);
            old.swap(idle_);
            idle_bytes_ = 0;
        }
        for (Block& b : old) free_block(b);
    }

  private:
    static void free_block(Block& b) {
        if (!b.ptr) return;
        if (b.from.release) b.from.release(b.ptr, b.from.user);
        else if (!b.from.alloc) std::free(b.ptr);
        b.ptr = nullptr;
    }
    static size_t limit() {
        static const size_t bytes = [] {
            const char* mb = std::getenv("DEEPBINNER_FAST5_POOL_MB");
            const long v = mb ? std::atol(mb) : 2048;
            return (size_t)(v < 0 ? 0 : v) << 20;
        }();
        return bytes;
    }
    std::mutex m_;
    std::vector<Block> idle_;
    size_t idle_bytes_ = 0;
    SampleAllocator allocator_;
};

SamplePool& sample_pool() {
    static SamplePool* pool = new SamplePool;      // never destroyed: batches may outlive exit()
    return *pool;
}

struct SampleBuffer {
    SamplePool::Block block;
    size_t count = 0;
    SampleBuffer() = default;
    SampleBuffer(const SampleBuffer&) = delete;
    SampleBuffer& operator=(const SampleBuffer&) = delete;
    ~SampleBuffer() { sample_pool().give(block); }
    void resize(size_t n) {
        sample_pool().give(block);
        block = SamplePool::Block();
        block = sample_pool().take(n * sizeof(int16_t));
        count = n;
    }
    int16_t* data() const { return static_cast<int16_t*>(block.ptr); }
};

struct f5_batch {
    SampleBuffer samples;
    std::vector<int64_t> offsets;
    std::vector<int32_t> status;
    std::vector<char> read_ids;
    // raw batches (f5_stream_open_raw): the Signal pieces as stored, for a decoder elsewhere
    SampleBuffer comp;                       // bytes (its count is in int16 units: see comp_bytes)
    int64_t comp_bytes = 0;
    std::vector<f5_raw_stream> streams;
};

extern "C" {

const char* f5_version(void) { return "deepbinner_fast5 0.1"; }

int f5_usable_cpus(void) { return usable_cpus(); }

const char* f5_status_string(int status) {
    switch (status) {
        case F5_OK: return "ok";
        case F5_ERR_OPEN: return "cannot open file";
        case F5_ERR_FORMAT: return "not a readable HDF5/fast5 file";
        case F5_ERR_NO_READ: return "no such read (or read without read_id / Signal)";
        case F5_ERR_MULTI: return "multi-read fast5 file";
        case F5_ERR_ARGUMENT: return "invalid argument";
        case F5_ERR_FILTER: return "Signal compressed with an unsupported filter (VBZ?)";
        case F5_ERR_EXISTS: return "a file of that name exists already (not overwritten)";
        default: return "unknown status";
    }
}

int f5_open(const char* path, f5_file** out) {
    if (!path || !out) return F5_ERR_ARGUMENT;
    *out = nullptr;
    f5_file* file = nullptr;
    const int rc = guarded([&] {
        file = new f5_file(path);
        file->impl.parse();
    });
    if (rc != F5_OK) {
        delete file;
        return rc;
    }
    *out = file;
    return F5_OK;
}

void f5_close(f5_file* file) { delete file; }

int f5_layout(f5_file* file, int* layout, int64_t* n_reads) {
    if (!file || !layout || !n_reads) return F5_ERR_ARGUMENT;
    *layout = file->impl.layout();
    *n_reads = file->impl.n_reads();
    return F5_OK;
}

int f5_read_info(f5_file* file, int64_t index, char read_id[F5_READ_ID_MAX], int64_t* n_samples) {
    if (!file || !read_id || !n_samples) return F5_ERR_ARGUMENT;
    return guarded([&] {
        const ReadEntry& r = file->impl.read(index);
        copy_read_id(r.read_id, read_id);
        *n_samples = r.signal.n;
    });
}

int f5_read_signal(f5_file* file, int64_t index, int64_t first, int64_t count, int16_t* out) {
    if (!file || (!out && count > 0)) return F5_ERR_ARGUMENT;
    return guarded([&] { file->impl.read_signal(file->impl.read(index).signal, first, count, out); });
}

int f5_load_batch(const char* const* paths, int64_t n_files, int64_t keep, int n_threads,
                  f5_batch** out) {
    if (!paths || !out || n_files < 0) return F5_ERR_ARGUMENT;
    *out = nullptr;
    f5_batch* batch = nullptr;
    try {
        batch = new f5_batch;
        batch->offsets.assign((size_t)n_files + 1, 0);
        batch->status.assign((size_t)n_files, F5_ERR_OPEN);
        batch->read_ids.assign((size_t)n_files * F5_READ_ID_MAX, 0);
        // Pass 1, per file and entirely on one worker thread: read the file into the thread's
        // buffer, parse it, inflate what is asked for into a staging block of its own.  Pass 2,
        // after a prefix sum of the lengths: copy the staging blocks into the packed buffer.
        // (Keeping the parsed files from a first pass to a second one cost an mmap-sized
        // allocation per file; the extra copy of 13-26 KB per read is nothing beside it.)
        std::vector<std::unique_ptr<int16_t[]>> staged((size_t)n_files);
        std::vector<int64_t> lengths((size_t)n_files, 0);

        auto load_one = [&](int64_t i) {
            thread_local std::vector<uint8_t> file_bytes;
            thread_local ChunkCache cache;
            cache.addr = ~0ull;            // what it holds is another file's chunk
            bool multi = false;
            int rc = paths[i] ? F5_OK : F5_ERR_ARGUMENT;
            if (rc == F5_OK) rc = guarded([&] {
                Fast5 file(paths[i], &file_bytes);
                file.parse();
                if (file.layout() == F5_LAYOUT_MULTI) {
                    multi = true;
                    return;
                }
                const ReadEntry& r = file.read(0);
                const int64_t n = r.signal.n;
                const int64_t kept = (keep > 0 && n > 2 * keep) ? 2 * keep : n;
                std::unique_ptr<int16_t[]> block(new int16_t[(size_t)std::max<int64_t>(kept, 1)]);
                if (kept < n) {
                    file.read_signal(r.signal, 0, keep, block.get(), &cache);
                    file.read_signal(r.signal, n - keep, keep, block.get() + keep, &cache);
                } else {
                    file.read_signal(r.signal, 0, n, block.get(), &cache);
                }
                copy_read_id(r.read_id, &batch->read_ids[(size_t)i * F5_READ_ID_MAX]);
                lengths[(size_t)i] = kept;
                staged[(size_t)i] = std::move(block);
            });
            if (multi) rc = F5_ERR_MULTI;
            if (rc != F5_OK) {      // unreadable, or damaged where only inflating shows it
                lengths[(size_t)i] = 0;
                std::memset(&batch->read_ids[(size_t)i * F5_READ_ID_MAX], 0, F5_READ_ID_MAX);
            }
            batch->status[(size_t)i] = rc;
        };
        auto pack_one = [&](int64_t i) {
            if (!staged[(size_t)i]) return;
            std::memcpy(batch->samples.data() + batch->offsets[(size_t)i], staged[(size_t)i].get(),
                        (size_t)lengths[(size_t)i] * 2);
            staged[(size_t)i].reset();
        };

        int threads = thread_count(n_threads);
        threads = (int)std::min<int64_t>(threads, std::max<int64_t>(n_files, 1));
        auto run_parallel = [&](const std::function<void(int64_t)>& fn) {
            worker_pool().run(threads, n_files, [&](int64_t i, int) { fn(i); });
        };

        run_parallel(load_one);
        int64_t total = 0;
        for (int64_t i = 0; i < n_files; ++i) {
            batch->offsets[(size_t)i] = total;
            total += lengths[(size_t)i];
        }
        batch->offsets[(size_t)n_files] = total;
        batch->samples.resize((size_t)total);
        run_parallel(pack_one);
    } catch (const std::exception&) {
        delete batch;
        return F5_ERR_OPEN;
    }
    *out = batch;
    return F5_OK;
}

// One-read files with their Signals AS STORED (the one-read twin of f5_stream_open_raw's
// batches): pass 1, per file on one worker thread, reads and parses the file and stages its
// Signal pieces' bytes; then the host's share of the inflating is chosen over the whole batch;
// pass 2 copies - or, for the host's share, inflates - the staged bytes into the batch's byte
// buffer.
int f5_load_batch_raw(const char* const* paths, int64_t n_files, int n_threads,
                      int64_t host_inflate_above, f5_batch** out) {
    if (!paths || !out || n_files < 0) return F5_ERR_ARGUMENT;
    *out = nullptr;
    f5_batch* batch = nullptr;
    try {
        batch = new f5_batch;
        batch->offsets.assign((size_t)n_files + 1, 0);
        batch->status.assign((size_t)n_files, F5_ERR_OPEN);
        batch->read_ids.assign((size_t)n_files * F5_READ_ID_MAX, 0);
        struct Staged {
            std::vector<Fast5::RawPiece> pieces;       // file_off = offset into `bytes` from here on
            std::string bytes;
            int64_t samples = 0;
        };
        std::vector<Staged> staged((size_t)n_files);
        const int64_t zlib_above = host_inflate_above > 0 ? host_inflate_above : 0;

        auto stage_one = [&](int64_t i) {
            thread_local std::vector<uint8_t> file_bytes;
            thread_local ChunkCache cache;
            cache.addr = ~0ull;
            bool multi = false;
            Staged& st = staged[(size_t)i];
            int rc = paths[i] ? F5_OK : F5_ERR_ARGUMENT;
            if (rc == F5_OK) rc = guarded([&] {
                Fast5 file(paths[i], &file_bytes);
                file.parse();
                if (file.layout() == F5_LAYOUT_MULTI) {
                    multi = true;
                    return;
                }
                const ReadEntry& r = file.read(0);
                st.samples = r.signal.n;
                file.signal_pieces(r.signal, zlib_above, &st.pieces);
                for (Fast5::RawPiece& p : st.pieces) {
                    const uint64_t wanted = (uint64_t)p.count * 2;
                    const size_t at = st.bytes.size();
                    if (p.kind == Fast5::kZlib || p.kind == Fast5::kStored) {
                        const uint64_t take = p.kind == Fast5::kZlib ? p.nbytes
                                                                     : std::min(p.nbytes, wanted);
                        st.bytes.resize(at + (size_t)take);
                        file.read_bytes(p.file_off, take, reinterpret_cast<uint8_t*>(&st.bytes[at]));
                        p.nbytes = take;
                    } else if (p.kind == Fast5::kHostDecode) {
                        st.bytes.resize(at + (size_t)wanted);
                        file.decode_piece(r.signal, p, reinterpret_cast<uint8_t*>(&st.bytes[at]),
                                          &cache);
                        p.kind = Fast5::kStored;
                        p.nbytes = wanted;
                    } else {
                        p.nbytes = 0;
                    }
                    p.file_off = at;
                }
                copy_read_id(r.read_id, &batch->read_ids[(size_t)i * F5_READ_ID_MAX]);
            });
            if (multi) rc = F5_ERR_MULTI;
            if (rc != F5_OK) {
                st = Staged();
                std::memset(&batch->read_ids[(size_t)i * F5_READ_ID_MAX], 0, F5_READ_ID_MAX);
            }
            batch->status[(size_t)i] = rc;
        };
        int threads = thread_count(n_threads);
        threads = (int)std::min<int64_t>(threads, std::max<int64_t>(n_files, 1));
        worker_pool().run(threads, n_files, [&](int64_t i, int) { stage_one(i); });

        // the host's share: the longest streams of the batch holding -host_inflate_above per cent
        // of its compressed bytes (kHostDecode from here on = "inflate the staged stream")
        if (host_inflate_above < 0) {
            std::vector<std::pair<uint64_t, std::pair<int64_t, size_t>>> streams;
            uint64_t total = 0;
            for (int64_t i = 0; i < n_files; ++i)
                for (size_t k = 0; k < staged[(size_t)i].pieces.size(); ++k)
                    if (staged[(size_t)i].pieces[k].kind == Fast5::kZlib) {
                        streams.push_back({staged[(size_t)i].pieces[k].nbytes, {i, k}});
                        total += staged[(size_t)i].pieces[k].nbytes;
                    }
            std::sort(streams.begin(), streams.end(),
                      [](const auto& a, const auto& b) { return a.first > b.first; });
            const uint64_t share = (uint64_t)std::min<int64_t>(-host_inflate_above, 100);
            uint64_t taken = 0;
            for (const auto& e : streams) {
                if (taken * 100 >= total * share) break;
                staged[(size_t)e.second.first].pieces[e.second.second].kind = Fast5::kHostDecode;
                taken += e.first;
            }
        }
        int64_t samples = 0, at = 0;
        size_t n_pieces = 0;
        for (int64_t i = 0; i < n_files; ++i) {
            batch->offsets[(size_t)i] = samples;
            samples += staged[(size_t)i].samples;
            n_pieces += staged[(size_t)i].pieces.size();
        }
        batch->offsets[(size_t)n_files] = samples;
        batch->streams.reserve(n_pieces);
        for (int64_t i = 0; i < n_files; ++i)
            for (Fast5::RawPiece& p : staged[(size_t)i].pieces) {
                p.comp_offset = at;
                p.comp_bytes = p.kind == Fast5::kHostDecode ? p.count * 2 : (int64_t)p.nbytes;
                at += p.comp_bytes;
                f5_raw_stream rec;
                rec.comp_offset = p.comp_offset;
                rec.comp_bytes = p.comp_bytes;
                rec.out_offset = (batch->offsets[(size_t)i] + p.first) * 2;
                rec.out_bytes = p.count * 2;
                rec.mode = p.kind == Fast5::kZlib ? F5_RAW_ZLIB : F5_RAW_STORED;
                rec.reserved = (int32_t)i;
                batch->streams.push_back(rec);
            }
        batch->comp_bytes = at;
        batch->comp.resize((size_t)(at + 64 + 1) / 2);
        uint8_t* comp = reinterpret_cast<uint8_t*>(batch->comp.data());
        std::memset(comp + at, 0, 64);
        std::vector<int32_t> inflate_failed((size_t)n_files, 0);
        worker_pool().run(threads, n_files, [&](int64_t i, int) {
            thread_local ChunkCache cache;
            thread_local std::vector<uint8_t> tmp;
            const Staged& st = staged[(size_t)i];
            for (const Fast5::RawPiece& p : st.pieces) {
                const uint8_t* src = reinterpret_cast<const uint8_t*>(st.bytes.data()) + p.file_off;
                uint8_t* dst = comp + p.comp_offset;
                if (p.kind != Fast5::kHostDecode) {
                    std::memcpy(dst, src, (size_t)p.comp_bytes);
                    continue;
                }
                const int rc = guarded([&] {
                    Fast5::inflate_stream(src, (size_t)p.nbytes, (size_t)p.count * 2 + 8, &tmp, &cache);
                });
                const size_t have = rc == F5_OK ? std::min(tmp.size(), (size_t)p.count * 2) : 0;
                if (have) std::memcpy(dst, tmp.data(), have);
                std::memset(dst + have, 0, (size_t)p.count * 2 - have);
                if (rc != F5_OK) inflate_failed[(size_t)i] = rc;
            }
        });
        for (int64_t i = 0; i < n_files; ++i)
            if (inflate_failed[(size_t)i]) {
                batch->status[(size_t)i] = inflate_failed[(size_t)i];
                std::memset(&batch->read_ids[(size_t)i * F5_READ_ID_MAX], 0, F5_READ_ID_MAX);
            }
        std::stable_sort(batch->streams.begin(), batch->streams.end(),
                         [](const f5_raw_stream& x, const f5_raw_stream& y) {
                             const int64_t wx = x.mode == F5_RAW_ZLIB ? x.comp_bytes : 0;
                             const int64_t wy = y.mode == F5_RAW_ZLIB ? y.comp_bytes : 0;
                             return wx > wy;
                         });
    } catch (const std::exception&) {
        delete batch;
        return F5_ERR_OPEN;
    }
    *out = batch;
    return F5_OK;
}

int f5_load_reads(const char* path, int64_t first, int64_t count, int64_t keep, int n_threads,
                  f5_batch** out) {
    if (!path || !out || first < 0) return F5_ERR_ARGUMENT;
    *out = nullptr;
    f5_batch* batch = nullptr;
    try {
        // ONE reader for all worker threads: the file is opened and its root group parsed once;
        // pass 1 resolves the reads (every read by exactly one thread, each touching only its
        // own entry), after which the reader is read-only and pass 2 needs per-thread state only
        // for the chunk inflated last.  Two passes like f5_load_batch: lengths, prefix sum,
        // inflate into place.
        const bool timing = std::getenv("DEEPBINNER_FAST5_TIMING") != nullptr;
        auto now = [] { return std::chrono::steady_clock::now(); };
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
            return std::chrono::duration<double, std::milli>(b - a).count();
        };
        const auto t_start = now();
        std::unique_ptr<Fast5> shared;
        const int open_status = guarded([&] {
            shared.reset(new Fast5(path));
            shared->parse();
        });
        if (open_status != F5_OK) return open_status;
        if (count < 0) count = std::max<int64_t>(shared->n_reads() - first, 0);   // to the end
        if (first + count > shared->n_reads()) return F5_ERR_NO_READ;
        int threads = thread_count(n_threads);
        threads = (int)std::min<int64_t>(threads, std::max<int64_t>(count, 1));
        std::vector<ChunkCache> caches((size_t)threads);
        const auto t_parsed = now();

        batch = new f5_batch;
        batch->offsets.assign((size_t)count + 1, 0);
        batch->status.assign((size_t)count, F5_ERR_OPEN);
        batch->read_ids.assign((size_t)count * F5_READ_ID_MAX, 0);
        std::vector<int64_t> lengths((size_t)count, 0);

        auto run_parallel = [&](const std::function<void(Fast5&, int64_t, ChunkCache*)>& fn) {
            worker_pool().run(threads, count, [&](int64_t i, int slot) {
                fn(*shared, i, &caches[(size_t)slot]);
            });
        };
        run_parallel([&](Fast5& file, int64_t i, ChunkCache*) {
            batch->status[(size_t)i] = guarded([&] {
                const ReadEntry& r = file.read(first + i);
                const int64_t n = r.signal.n;
                lengths[(size_t)i] = (keep > 0 && n > 2 * keep) ? 2 * keep : n;
                copy_read_id(r.read_id, &batch->read_ids[(size_t)i * F5_READ_ID_MAX]);
            });
        });
        const auto t_resolved = now();
        int64_t total = 0;
        for (int64_t i = 0; i < count; ++i) {
            batch->offsets[(size_t)i] = total;
            total += batch->status[(size_t)i] == F5_OK ? lengths[(size_t)i] : 0;
        }
        batch->offsets[(size_t)count] = total;
        batch->samples.resize((size_t)total);
        const auto t_allocated = now();
        run_parallel([&](Fast5& file, int64_t i, ChunkCache* cache) {
            if (batch->status[(size_t)i] != F5_OK) return;
            int16_t* dst = batch->samples.data() + batch->offsets[(size_t)i];
            const int rc = guarded([&] {
                const ReadEntry& r = file.read(first + i);
                const int64_t n = r.signal.n;
                if (keep > 0 && n > 2 * keep) {
                    file.read_signal(r.signal, 0, keep, dst, cache);
                    file.read_signal(r.signal, n - keep, keep, dst + keep, cache);
                } else {
                    file.read_signal(r.signal, 0, n, dst, cache);
                }
            });
            if (rc != F5_OK) std::memset(dst, 0, (size_t)lengths[(size_t)i] * 2);
            batch->status[(size_t)i] = rc;
        });
        for (int64_t i = 0; i < count; ++i)
            if (batch->status[(size_t)i] != F5_OK)
                std::memset(&batch->read_ids[(size_t)i * F5_READ_ID_MAX], 0, F5_READ_ID_MAX);
        if (timing)
            std::fprintf(stderr,
                         "f5_load_reads: %lld reads, %d threads: open+parse %.1f ms, resolve %.1f ms, "
                         "allocate %.1f ms, inflate %.1f ms\n",
                         (long long)count, threads, ms(t_start, t_parsed), ms(t_parsed, t_resolved),
                         ms(t_resolved, t_allocated), ms(t_allocated, now()));
    } catch (const std::exception&) {
        delete batch;
        return F5_ERR_OPEN;
    }
    *out = batch;
    return F5_OK;
}

int f5_write_single_reads(const char* container, int64_t n, const int64_t* read_index,
                          const char* const* out_paths, int n_threads, int32_t* status,
                          int64_t* bytes_written) {
    if (bytes_written) *bytes_written = 0;
    if (!container || n < 0 || (n > 0 && (!read_index || !out_paths || !status)))
        return F5_ERR_ARGUMENT;
    if (n == 0) return F5_OK;
    try {
        std::unique_ptr<Fast5> shared;
        const int open_status = guarded([&] {
            shared.reset(new Fast5(container));
            shared->parse();
        });
        if (open_status != F5_OK) return open_status;
        // pass 1: every wanted read of the container resolved by exactly one thread (a read asked
        // for twice must not be resolved by two at once); from then on the reader is read-only
        std::vector<int64_t> wanted(read_index, read_index + n);
        std::sort(wanted.begin(), wanted.end());
        wanted.erase(std::unique(wanted.begin(), wanted.end()), wanted.end());
        int threads = thread_count(n_threads);
        threads = (int)std::min<int64_t>(threads, n);
        worker_pool().run(threads, (int64_t)wanted.size(), [&](int64_t k, int) {
            (void)guarded([&] { (void)shared->read(wanted[(size_t)k]); });
        });
        std::vector<ChunkCache> caches((size_t)threads);
        std::atomic<int64_t> written(0);
        worker_pool().run(threads, n, [&](int64_t i, int slot) {
            status[i] = guarded([&] {
                if (!out_paths[i]) throw std::out_of_range("no path");
                Fast5& file = *shared;
                const ReadEntry& r = file.read(read_index[i]);      // resolved: read-only now
                if (!r.resolved) throw std::out_of_range("read not resolved");
                Fast5::ReadMeta meta;
                try {
                    file.read_metadata(read_index[i], &meta);
                } catch (const std::exception&) {
                    meta = Fast5::ReadMeta();      // the signal and the read id are what binning
                }                                  // cannot do without; the rest is left behind
                const int64_t samples = r.signal.n;
                std::string packed;
                uint64_t off = 0, nbytes = 0;
                if (samples > 0 && file.whole_deflated_chunk(r.signal, &off, &nbytes)) {
                    packed.resize((size_t)nbytes);
                    file.read_bytes(off, nbytes, reinterpret_cast<uint8_t*>(&packed[0]));
                } else if (samples > 0) {
                    // stored some other way (contiguous, several chunks, more filters): decoded and
                    // deflated again, level 1 like hdf5_write.py
                    std::vector<int16_t> raw((size_t)samples);
                    file.read_signal(r.signal, 0, samples, raw.data(), &caches[(size_t)slot]);
                    uLongf bound = compressBound((uLong)samples * 2);
                    packed.resize((size_t)bound);
                    if (compress2(reinterpret_cast<Bytef*>(&packed[0]), &bound,
                                  reinterpret_cast<const Bytef*>(raw.data()), (uLong)samples * 2,
                                  1) != Z_OK)
                        throw FormatError("deflate failed");
                    packed.resize((size_t)bound);
                }
                const std::string image =
                    h5w::single_read_file(r.read_id, (uint64_t)samples, packed, meta);
                // Never over a file that is there (the reference's move counts a clash and leaves
                // the earlier file alone, realtime.py:111-144), never through a symlink, and no
                // half-written file under the final name: the image goes to a fresh temporary
                // name in the same directory and is linked into place - link() fails if the name
                // exists, and is atomic.
                if (::access(out_paths[i], F_OK) == 0) throw ExistsError();
                const std::string tmp = std::string(out_paths[i]) + ".part." +
                                        std::to_string((long long)::getpid()) + "." +
                                        std::to_string((long long)i);
                ::unlink(tmp.c_str());      // (a stale one of this very name: a killed run whose
                                            // pid came round - the O_EXCL below would refuse it)
                write_exclusive(tmp.c_str(), image);
                int linked = -1, link_errno = EPERM;
                if (!links_forbidden()) {
                    linked = ::link(tmp.c_str(), out_paths[i]);
                    link_errno = errno;
                }
                if (linked != 0 && link_errno != EEXIST && no_hard_links(link_errno)) {
                    // exFAT / FAT, many SMB and FUSE mounts (sequencing drives) have no link(): the
                    // finished temporary file is RENAMED into place - nobody ever sees a partial
                    // file under the final name, and a killed process leaves only the .part file -
                    // without replacing (renameat2 RENAME_NOREPLACE); where the filesystem does not
                    // know that flag, after a look at the name (rename is atomic there too; the
                    // window between the look and the rename is the one thing link() had closed)
                    int renamed = -1;
#ifdef RENAME_NOREPLACE
                    renamed = (int)::syscall(SYS_renameat2, AT_FDCWD, tmp.c_str(), AT_FDCWD, out_paths[i],
                                             RENAME_NOREPLACE);
                    if (renamed != 0 && errno == EEXIST) {
                        ::unlink(tmp.c_str());
                        throw ExistsError();
                    }
#endif
                    if (renamed != 0) {
                        struct stat st_there;
                        if (::lstat(out_paths[i], &st_there) == 0) {
                            ::unlink(tmp.c_str());
                            throw ExistsError();
                        }
                        if (::rename(tmp.c_str(), out_paths[i]) != 0) {
                            ::unlink(tmp.c_str());
                            throw std::runtime_error("cannot name file");
                        }
                    }
                } else {
                    ::unlink(tmp.c_str());
                    if (linked != 0) {
                        if (link_errno == EEXIST) throw ExistsError();
                        throw std::runtime_error("cannot name file");
                    }
                }
                written.fetch_add((int64_t)image.size());
            });
        });
        if (bytes_written) *bytes_written = written.load();
    } catch (const std::exception&) {
        return F5_ERR_OPEN;
    }
    return F5_OK;
}

/* the same file as bytes (tests: held against hdf5_write.single_read_fast5_bytes) */
int f5_single_read_image(const char* container, int64_t read_index, uint8_t* out, int64_t capacity,
                         int64_t* size) {
    if (!container || !size || capacity < 0 || (capacity > 0 && !out)) return F5_ERR_ARGUMENT;
    *size = 0;
    return guarded([&] {
        Fast5 file(container);
        file.parse();
        const ReadEntry& r = file.read(read_index);
        Fast5::ReadMeta meta;
        file.read_metadata(read_index, &meta);
        std::string packed;
        uint64_t off = 0, nbytes = 0;
        if (r.signal.n > 0 && file.whole_deflated_chunk(r.signal, &off, &nbytes)) {
            packed.resize((size_t)nbytes);
            file.read_bytes(off, nbytes, reinterpret_cast<uint8_t*>(&packed[0]));
        } else if (r.signal.n > 0) {
            std::vector<int16_t> raw((size_t)r.signal.n);
            file.read_signal(r.signal, 0, r.signal.n, raw.data());
            uLongf bound = compressBound((uLong)r.signal.n * 2);
            packed.resize((size_t)bound);
            if (compress2(reinterpret_cast<Bytef*>(&packed[0]), &bound,
                          reinterpret_cast<const Bytef*>(raw.data()), (uLong)r.signal.n * 2, 1) != Z_OK)
                throw FormatError("deflate failed");
            packed.resize((size_t)bound);
        }
        const std::string image = h5w::single_read_file(r.read_id, (uint64_t)r.signal.n, packed, meta);
        *size = (int64_t)image.size();
        if ((int64_t)image.size() <= capacity) std::memcpy(out, image.data(), image.size());
    });
}

int64_t f5_batch_size(const f5_batch* batch) { return batch ? (int64_t)batch->status.size() : 0; }
const int16_t* f5_batch_samples(const f5_batch* batch) { return batch ? batch->samples.data() : nullptr; }
const int64_t* f5_batch_offsets(const f5_batch* batch) { return batch ? batch->offsets.data() : nullptr; }
const int32_t* f5_batch_status(const f5_batch* batch) { return batch ? batch->status.data() : nullptr; }
const char* f5_batch_read_ids(const f5_batch* batch) { return batch ? batch->read_ids.data() : nullptr; }
void f5_batch_free(f5_batch* batch) { delete batch; }

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// f5_stream: multi-read containers in, packed batches out, SEVERAL containers in flight.
// One f5_load_reads call per container leaves most of a many-core host idle most of the time:
// opening a 4,000-read container and walking its root group is 6-10 ms on ONE thread, the reads
// are resolved and inflated in two passes with a join after each, and the caller's own work per
// batch happens while nothing loads.  Here a team of threads works on a window of `depth`
// containers at once, always taking the oldest work there is: read ranges of a container that is
// being inflated, else ranges of one being resolved, else a container that still has to be opened -
// so that the next containers' serial parts run beside the current one's inflating, and the batches
// still come out in path order.
// ---------------------------------------------------------------------------------------------
struct f5_stream {
    enum Phase { kNew, kParsing, kResolve, kLayout, kInflate, kDone };
    struct Container {
        int64_t index = 0;
        std::string path;
        std::unique_ptr<Fast5> file;
        int status = F5_OK;                 // why the container could not be opened / laid out
        Phase phase = kNew;
        int64_t count = 0, next = 0, done = 0;
        f5_batch* batch = nullptr;
        std::vector<int64_t> lengths;
        std::vector<std::vector<Fast5::RawPiece>> pieces;      // raw mode: per read
        std::vector<int64_t> fetch_order;   // raw mode: the reads by where their Signal lies in the file
        std::chrono::steady_clock::time_point stamp[5];        // DEEPBINNER_FAST5_TIMING
        ~Container() { delete batch; }
    };
    struct Task {
        Container* c = nullptr;
        Phase phase = kNew;
        int64_t a = 0, b = 0;
    };

    std::vector<std::string> paths;
    int64_t keep = 0;
    int depth = 3;
    bool raw = false;                // hand out the Signal pieces as stored (f5_stream_open_raw)
    int64_t zlib_above = 0;          // ... except deflate streams longer than this: host-inflated
    int host_share = 0;              // ... and the longest ones holding this share (%) of the bytes
    static constexpr int64_t kLongStreamBytes = 64 * 1024;
    std::mutex m;
    std::condition_variable work_cv, done_cv;
    std::deque<std::unique_ptr<Container>> inflight;      // in path order
    int64_t next_path = 0;
    bool stop = false;
    std::vector<std::thread> workers;

    static constexpr int64_t kInflateGrain = 8, kResolveGrain = 64, kFetchGrain = 256;

    ~f5_stream() {
        {
            std::lock_guard<std::mutex> g(m);
            stop = true;
        }
        work_cv.notify_all();
        for (std::thread& t : workers) t.join();
    }

    void admit_locked() {
        while ((int)inflight.size() < depth && next_path < (int64_t)paths.size()) {
            std::unique_ptr<Container> c(new Container);
            c->index = next_path;
            c->path = paths[(size_t)next_path++];
            inflight.push_back(std::move(c));
        }
    }

    bool pick_locked(Task* t) {
        for (auto& up : inflight) {
            Container* c = up.get();
            if ((c->phase == kInflate || c->phase == kResolve) && c->next < c->count) {
                const int64_t grain = c->phase != kInflate ? kResolveGrain : raw ? kFetchGrain : kInflateGrain;
                t->c = c;
                t->phase = c->phase;
                t->a = c->next;
                t->b = std::min(c->count, c->next + grain);
                c->next = t->b;
                return true;
            }
        }
        for (auto& up : inflight) {
            Container* c = up.get();
            if (c->phase == kNew) {
                c->phase = kParsing;
                t->c = c;
                t->phase = kParsing;
                return true;
            }
        }
        return false;
    }

    static bool timing() {
        static const bool on = std::getenv("DEEPBINNER_FAST5_TIMING") != nullptr;
        return on;
    }
    static void mark(Container* c, int k) {
        if (timing()) c->stamp[k] = std::chrono::steady_clock::now();
    }
    static void report(const Container* c) {
        if (!timing()) return;
        auto ms = [&](int a, int b) {
            return std::chrono::duration<double, std::milli>(c->stamp[b] - c->stamp[a]).count();
        };
        std::fprintf(stderr,
                     "f5_stream: container %lld, %lld reads: open+parse %.1f ms, resolve %.1f ms, "
                     "layout %.1f ms, %s %.1f ms\n",
                     (long long)c->index, (long long)c->count, ms(0, 1), ms(1, 2), ms(2, 3),
                     c->pieces.empty() ? "inflate" : "fetch", ms(3, 4));
    }

    void parse(Container* c) {
        mark(c, 0);
        c->status = guarded([&] {
            c->file.reset(new Fast5(c->path.c_str()));
            c->file->parse();
        });
        if (c->status == F5_OK) {
            c->count = c->file->n_reads();
            try {
                c->batch = new f5_batch;
                c->batch->offsets.assign((size_t)c->count + 1, 0);
                c->batch->status.assign((size_t)c->count, F5_ERR_OPEN);
                c->batch->read_ids.assign((size_t)c->count * F5_READ_ID_MAX, 0);
                c->lengths.assign((size_t)c->count, 0);
                if (raw) c->pieces.resize((size_t)c->count);
            } catch (const std::exception&) {
                c->status = F5_ERR_FORMAT;
            }
        }
        mark(c, 1);
    }

    void resolve(Container* c, int64_t a, int64_t b) {
        for (int64_t i = a; i < b; ++i)
            c->batch->status[(size_t)i] = guarded([&] {
                const ReadEntry& r = c->file->read(i);
                const int64_t n = r.signal.n;
                c->lengths[(size_t)i] = (!raw && keep > 0 && n > 2 * keep) ? 2 * keep : n;
                if (raw) c->file->signal_pieces(r.signal, zlib_above, &c->pieces[(size_t)i]);
                copy_read_id(r.read_id, &c->batch->read_ids[(size_t)i * F5_READ_ID_MAX]);
            });
    }

    void layout(Container* c) {
        mark(c, 2);
        struct MarkEnd {
            Container* c;
            ~MarkEnd() { mark(c, 3); }
        } mark_end{c};
        int64_t total = 0;
        for (int64_t i = 0; i < c->count; ++i) {
            c->batch->offsets[(size_t)i] = total;
            total += c->batch->status[(size_t)i] == F5_OK ? c->lengths[(size_t)i] : 0;
        }
        c->batch->offsets[(size_t)c->count] = total;
        try {
            if (!raw) {
                c->batch->samples.resize((size_t)total);
                return;
            }
            // A share of the inflating for the host: with host_share > 0 the longest deflate streams
            // holding that share (per cent) of the container's compressed bytes are left to the
            // host's threads - long streams are what a CPU core is good at (a stream is one
            // lane's work on the GPU however long it is) and what is left for the GPU is of even
            // length.
            if (host_share > 0) {
                std::vector<int64_t> sizes;
                int64_t all = 0;
                for (int64_t i = 0; i < c->count; ++i)
                    if (c->batch->status[(size_t)i] == F5_OK)
                        for (const Fast5::RawPiece& p : c->pieces[(size_t)i])
                            if (p.kind == Fast5::kZlib) {
                                sizes.push_back((int64_t)p.nbytes);
                                all += (int64_t)p.nbytes;
                            }
                std::sort(sizes.begin(), sizes.end(), std::greater<int64_t>());
                int64_t cut = INT64_MAX, taken = 0;
                // ... of a container of ORDINARY reads.  Where the streams are long throughout
                // (mean above kLongStreamBytes: reads of ~50 k samples and more) the host keeps
                // none: a 200 KB stream is 0.6 ms of a core, and with a wavefront per stream the GPU
                // decodes such containers alone at 134 k reads/s where a fifth of the bytes on the
                // host's sixteen threads made it 54 k (profiles/r06_loader).
                const bool long_streams = !sizes.empty() && all / (int64_t)sizes.size() > kLongStreamBytes;
                for (int64_t v : sizes) {
                    if (long_streams || taken * 100 >= all * host_share) break;
                    taken += v;
                    cut = v;
                }
                for (int64_t i = 0; i < c->count; ++i)
                    if (c->batch->status[(size_t)i] == F5_OK)
                        for (Fast5::RawPiece& p : c->pieces[(size_t)i])
                            if (p.kind == Fast5::kZlib && (int64_t)p.nbytes >= cut)
                                p.kind = Fast5::kHostDecode;
            }
            // raw: every piece gets its place in the byte buffer (what it occupies there: the
            // stored bytes, or - decoded by the host - its samples) and its record; the records
            // go out longest stream first, so that the lanes of a wave of the GPU decoder - which
            // step together, one stream each - get streams of one length
            int64_t at = 0;
            size_t n_pieces = 0;
            for (int64_t i = 0; i < c->count; ++i)
                if (c->batch->status[(size_t)i] == F5_OK) n_pieces += c->pieces[(size_t)i].size();
            c->batch->streams.reserve(n_pieces);
            for (int64_t i = 0; i < c->count; ++i) {
                if (c->batch->status[(size_t)i] != F5_OK) continue;
                for (Fast5::RawPiece& p : c->pieces[(size_t)i]) {
                    const int64_t wanted = p.count * 2;
                    p.comp_offset = at;
                    p.comp_bytes = p.kind == Fast5::kZlib     ? (int64_t)p.nbytes
                                   : p.kind == Fast5::kStored ? std::min<int64_t>((int64_t)p.nbytes, wanted)
                                   : p.kind == Fast5::kZeros  ? 0
                                                              : wanted;
                    at += p.comp_bytes;
                    f5_raw_stream rec;
                    rec.comp_offset = p.comp_offset;
                    rec.comp_bytes = p.comp_bytes;
                    rec.out_offset = (c->batch->offsets[(size_t)i] + p.first) * 2;
                    rec.out_bytes = wanted;
                    rec.mode = p.kind == Fast5::kZlib ? F5_RAW_ZLIB : F5_RAW_STORED;
                    rec.reserved = (int32_t)i;         // which read of the batch it belongs to
                    c->batch->streams.push_back(rec);
                }
            }
            std::stable_sort(c->batch->streams.begin(), c->batch->streams.end(),
                             [](const f5_raw_stream& x, const f5_raw_stream& y) {
                                 const int64_t wx = x.mode == F5_RAW_ZLIB ? x.comp_bytes : 0;
                                 const int64_t wy = y.mode == F5_RAW_ZLIB ? y.comp_bytes : 0;
                                 return wx > wy;
                             });
            c->batch->comp_bytes = at;
            // the fetch pass walks the reads in FILE order (groups are listed by name - random
            // read ids - but written one after the other): neighbours in a task are neighbours in
            // the file, and Fast5::read_many reads them together
            c->fetch_order.resize((size_t)c->count);
            for (int64_t i = 0; i < c->count; ++i) c->fetch_order[(size_t)i] = i;
            {
                std::vector<uint64_t> where((size_t)c->count, ~0ull);
                for (int64_t i = 0; i < c->count; ++i)
                    if (c->batch->status[(size_t)i] == F5_OK)
                        for (const Fast5::RawPiece& p : c->pieces[(size_t)i])
                            if (p.kind == Fast5::kZlib || p.kind == Fast5::kStored) {
                                where[(size_t)i] = p.file_off;
                                break;
                            }
                std::sort(c->fetch_order.begin(), c->fetch_order.end(),
                          [&](int64_t x, int64_t y) { return where[(size_t)x] < where[(size_t)y]; });
            }
            // (the decoder fetches ahead of itself: 64 readable bytes behind the last stream)
            c->batch->comp.resize((size_t)(at + 64 + 1) / 2);
            std::memset(reinterpret_cast<uint8_t*>(c->batch->comp.data()) + at, 0, 64);
        } catch (const std::exception&) {
            c->status = F5_ERR_FORMAT;
        }
    }

    // raw mode's pass 2: the pieces of the reads at places [a, b) of the container's fetch order
    // into the byte buffer - the stored ones together (Fast5::read_many), then what the host decodes
    void fetch(Container* c, int64_t a, int64_t b) {
        thread_local ChunkCache cache;
        thread_local std::vector<Fast5::IoItem> items;
        thread_local std::vector<char> failed;
        uint8_t* comp = reinterpret_cast<uint8_t*>(c->batch->comp.data());
        auto fail = [&](int64_t i, int rc) {
            // what could not be fetched or decoded reads as nothing: a stored stream of no
            // bytes is zero-extended by the decoder; the read is marked
            for (f5_raw_stream& rec : c->batch->streams)
                if (rec.reserved == (int32_t)i) {
                    rec.mode = F5_RAW_STORED;
                    rec.comp_bytes = 0;
                }
            std::memset(&c->batch->read_ids[(size_t)i * F5_READ_ID_MAX], 0, F5_READ_ID_MAX);
            c->batch->status[(size_t)i] = rc;
        };
        items.clear();
        for (int64_t k = a; k < b; ++k) {
            const int64_t i = c->fetch_order[(size_t)k];
            if (c->batch->status[(size_t)i] != F5_OK) continue;
            for (const Fast5::RawPiece& p : c->pieces[(size_t)i])
                if (p.kind == Fast5::kZlib || p.kind == Fast5::kStored)
                    items.push_back({p.file_off, (uint64_t)p.comp_bytes, comp + p.comp_offset, i});
        }
        const int rc_all = guarded([&] { c->file->read_many(items, &failed); });
        for (size_t k = 0; k < items.size(); ++k)
            if ((rc_all != F5_OK || failed[k]) && c->batch->status[(size_t)items[k].tag] == F5_OK)
                fail(items[k].tag, rc_all != F5_OK ? rc_all : F5_ERR_FORMAT);
        for (int64_t k = a; k < b; ++k) {
            const int64_t i = c->fetch_order[(size_t)k];
            if (c->batch->status[(size_t)i] != F5_OK) continue;
            const int rc = guarded([&] {
                const ReadEntry& r = c->file->read(i);
                for (const Fast5::RawPiece& p : c->pieces[(size_t)i])
                    if (p.kind == Fast5::kHostDecode)
                        c->file->decode_piece(r.signal, p, comp + p.comp_offset, &cache);
            });
            if (rc != F5_OK) fail(i, rc);
        }
    }

    void inflate(Container* c, int64_t a, int64_t b) {
        thread_local ChunkCache cache;
        for (int64_t i = a; i < b; ++i) {
            if (c->batch->status[(size_t)i] != F5_OK) continue;
            int16_t* dst = c->batch->samples.data() + c->batch->offsets[(size_t)i];
            const int rc = guarded([&] {
                const ReadEntry& r = c->file->read(i);
                const int64_t n = r.signal.n;
                if (keep > 0 && n > 2 * keep) {
                    c->file->read_signal(r.signal, 0, keep, dst, &cache);
                    c->file->read_signal(r.signal, n - keep, keep, dst + keep, &cache);
                } else {
                    c->file->read_signal(r.signal, 0, n, dst, &cache);
                }
            });
            if (rc != F5_OK) {
                std::memset(dst, 0, (size_t)c->lengths[(size_t)i] * 2);
                std::memset(&c->batch->read_ids[(size_t)i * F5_READ_ID_MAX], 0, F5_READ_ID_MAX);
            }
            c->batch->status[(size_t)i] = rc;
        }
    }

    void finish(Container* c) {      // no read of it is being worked on any more
        mark(c, 4);
        report(c);
        for (int64_t i = 0; i < c->count; ++i)
            if (c->batch->status[(size_t)i] != F5_OK)
                std::memset(&c->batch->read_ids[(size_t)i * F5_READ_ID_MAX], 0, F5_READ_ID_MAX);
        c->file.reset();             // unmap and close now, not when the caller frees the batch
    }

    void worker() {
        // (a name an operator - and tools/gpu_inflate_split.py's CPU accounting - can tell from the
        // interpreter's and the GPU runtime's threads)
        pthread_setname_np(pthread_self(), "f5-stream");
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            Task t;
            while (!stop && !pick_locked(&t)) work_cv.wait(lk);
            if (stop) return;
            Container* c = t.c;
            lk.unlock();
            if (t.phase == kParsing) parse(c);
            else if (t.phase == kResolve) resolve(c, t.a, t.b);
            else if (raw) fetch(c, t.a, t.b);
            else inflate(c, t.a, t.b);
            lk.lock();
            if (t.phase == kParsing) {
                if (c->status != F5_OK) {
                    c->phase = kDone;
                    done_cv.notify_all();
                    continue;
                }
                c->next = c->done = 0;
                c->phase = c->count > 0 ? kResolve : kLayout;
                if (c->count > 0) {
                    work_cv.notify_all();
                    continue;
                }
            } else {
                c->done += t.b - t.a;
                if (c->done < c->count) continue;
                if (t.phase == kInflate) {
                    lk.unlock();
                    finish(c);
                    lk.lock();
                    c->phase = kDone;
                    done_cv.notify_all();
                    continue;
                }
                c->phase = kLayout;
            }
            // the last range of the resolve pass (or an empty container): sizes are known
            lk.unlock();
            layout(c);
            if (c->status == F5_OK && c->count == 0) finish(c);
            lk.lock();
            if (c->status != F5_OK || c->count == 0) {
                c->phase = kDone;
                done_cv.notify_all();
            } else {
                c->next = c->done = 0;
                c->phase = kInflate;
                work_cv.notify_all();
            }
        }
    }
};

extern "C" {

int f5_set_sample_allocator(f5_alloc_fn alloc, f5_free_fn release, void* user) {
    if ((alloc == nullptr) != (release == nullptr)) return F5_ERR_ARGUMENT;
    sample_pool().set_allocator(alloc, release, user);
    return F5_OK;
}

void f5_release_idle_buffers(void) { sample_pool().flush(); }

int f5_stream_open(const char* const* paths, int64_t n_paths, int64_t keep, int n_threads,
                   int depth, f5_stream** out) {
    if (!out || n_paths < 0 || (n_paths > 0 && !paths)) return F5_ERR_ARGUMENT;
    *out = nullptr;
    for (int64_t i = 0; i < n_paths; ++i)
        if (!paths[i]) return F5_ERR_ARGUMENT;
    f5_stream* s = nullptr;
    try {
        s = new f5_stream;
        s->paths.assign(paths, paths + n_paths);
        s->keep = keep;
        s->depth = depth > 0 ? std::min(depth, 64) : 3;
        {
            std::lock_guard<std::mutex> g(s->m);
            s->admit_locked();
        }
        const int threads = thread_count(n_threads);
        for (int t = 0; t < threads; ++t) s->workers.emplace_back([s] { s->worker(); });
    } catch (const std::exception&) {
        delete s;
        return F5_ERR_OPEN;
    }
    *out = s;
    return F5_OK;
}

int f5_stream_open_raw(const char* const* paths, int64_t n_paths, int n_threads, int depth,
                       int64_t host_inflate_above, f5_stream** out) {
    f5_stream* s = nullptr;
    // A raw container is little work per read (no inflating) behind a serial start (one thread
    // opens and parses it: 5-6 ms of the ~30 ms of CPU a container of 4,000 reads costs): a window
    // of three starves a team of sixteen - 1.2 M reads/s where eight in flight give 1.7 M and
    // sixteen 1.9 M (profiles/r06_loader).  Default: half the team, between 3 and 8.
    if (depth <= 0) depth = std::max(3, std::min(8, thread_count(n_threads) / 2));
    // (opened without a team first, so that the mode is set before any thread looks at it)
    const int st = f5_stream_open(paths, 0, 0, 1, depth, &s);
    if (st != F5_OK) return st;
    if (!out || n_paths < 0 || (n_paths > 0 && !paths)) {
        delete s;
        return F5_ERR_ARGUMENT;
    }
    try {
        for (int64_t i = 0; i < n_paths; ++i) {
            if (!paths[i]) {
                delete s;
                return F5_ERR_ARGUMENT;
            }
            s->paths.emplace_back(paths[i]);
        }
        {
            std::lock_guard<std::mutex> g(s->m);
            s->raw = true;
            // (>= 0: a length in bytes; < 0: minus the host's share of the bytes in per cent)
            s->zlib_above = host_inflate_above > 0 ? host_inflate_above : 0;
            s->host_share = host_inflate_above < 0
                                ? (int)std::min<int64_t>(100, -host_inflate_above) : 0;
            s->admit_locked();
        }
        const int threads = thread_count(n_threads);
        for (int t = (int)s->workers.size(); t < threads; ++t)
            s->workers.emplace_back([s] { s->worker(); });
        s->work_cv.notify_all();
    } catch (const std::exception&) {
        delete s;
        return F5_ERR_OPEN;
    }
    *out = s;
    return F5_OK;
}

const uint8_t* f5_batch_comp(const f5_batch* batch) {
    return batch ? reinterpret_cast<const uint8_t*>(batch->comp.data()) : nullptr;
}
int64_t f5_batch_comp_bytes(const f5_batch* batch) { return batch ? batch->comp_bytes : 0; }
const f5_raw_stream* f5_batch_streams(const f5_batch* batch) {
    return batch && !batch->streams.empty() ? batch->streams.data() : nullptr;
}
int64_t f5_batch_n_streams(const f5_batch* batch) {
    return batch ? (int64_t)batch->streams.size() : 0;
}

int f5_stream_next(f5_stream* s, int64_t* index, int* container_status, f5_batch** batch) {
    if (!s || !index || !container_status || !batch) return F5_ERR_ARGUMENT;
    *batch = nullptr;
    std::unique_lock<std::mutex> lk(s->m);
    if (s->inflight.empty()) return F5_ERR_NO_READ;       // exhausted
    f5_stream::Container* front = s->inflight.front().get();
    s->done_cv.wait(lk, [&] { return front->phase == f5_stream::kDone; });
    std::unique_ptr<f5_stream::Container> c = std::move(s->inflight.front());
    s->inflight.pop_front();
    s->admit_locked();
    lk.unlock();
    s->work_cv.notify_all();
    *index = c->index;
    *container_status = c->status;
    if (c->status == F5_OK) {
        *batch = c->batch;
        c->batch = nullptr;
    }
    return F5_OK;
}

void f5_stream_close(f5_stream* s) { delete s; }

}  // extern "C"
