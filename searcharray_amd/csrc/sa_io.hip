// On-disk index <-> HBM.
//
// The reference persists an index as ONE raw file of uint64 roaringish words -- ArrayDict.data.tofile
// (searcharray/phrase/memmap_arrays.py:158-161) -- plus a metadata dict {term id: {offset, length}}
// in element units (memmap_arrays.py:28-54) that travels in the pickle; reading it back is a numpy
// memmap whose pages fault in term by term as queries touch them (memmap_arrays.py:163-165).
//
// Here the file is streamed straight into HBM: a reader thread preads 32 MiB pieces into a ring of
// page-locked buffers while the copy engine drains the previous pieces, so disk / page-cache reads and
// H2D copies overlap and no pageable host copy of the index ever exists.  Saving runs the same ring
// the other way (D2H -> pwrite), so an index encoded on the device reaches the disk without a host
// detour through numpy.
#include "sa_index.hpp"
#include "../../include/searcharray_hip.h"

#include <condition_variable>
#include <new>
#include <thread>

#include <errno.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {

constexpr size_t SA_IO_PIECE = 32u << 20;   // bytes per staged piece
constexpr int SA_IO_SLOTS = 3;              // ring depth: one being read, one in flight, one spare

struct Piece { u64 file_byte; u64 dev_word; u64 bytes; };

// Single-producer / single-consumer ring of pinned staging buffers.
struct Ring {
    void* buf[SA_IO_SLOTS] = {};
    hipEvent_t done[SA_IO_SLOTS] = {};      // device finished with the slot
    std::mutex mu;
    std::condition_variable cv;
    u64 produced = 0, consumed = 0;         // pieces filled by the io thread / released by the device side
    int io_errno = 0;
    bool abort = false;

    int init() {
        for (int i = 0; i < SA_IO_SLOTS; i++) {
            SA_HIP(hipHostMalloc(&buf[i], SA_IO_PIECE));
            SA_HIP(hipEventCreateWithFlags(&done[i], hipEventDisableTiming));
        }
        return SA_OK;
    }
    ~Ring() {
        for (int i = 0; i < SA_IO_SLOTS; i++) {
            if (done[i]) hipEventDestroy(done[i]);
            if (buf[i]) hipHostFree(buf[i]);
        }
    }
};

bool full_pread(int fd, void* dst, u64 bytes, u64 off) {
    char* p = (char*)dst;
    while (bytes) {
        const ssize_t r = pread(fd, p, bytes, (off_t)off);
        if (r < 0) { if (errno == EINTR) continue; return false; }
        if (r == 0) { errno = ENODATA; return false; }          // file shorter than the metadata says
        p += r; off += (u64)r; bytes -= (u64)r;
    }
    return true;
}

bool full_pwrite(int fd, const void* src, u64 bytes, u64 off) {
    const char* p = (const char*)src;
    while (bytes) {
        const ssize_t r = pwrite(fd, p, bytes, (off_t)off);
        if (r < 0) { if (errno == EINTR) continue; return false; }
        p += r; off += (u64)r; bytes -= (u64)r;
    }
    return true;
}

// file -> device: the io thread fills slots, this thread issues the copies
int stream_in(int fd, const std::vector<Piece>& pieces, u64* d_words, hipStream_t st) {
    Ring ring;
    SA_TRY(ring.init());
    std::thread io([&] {
        for (u64 k = 0; k < pieces.size(); k++) {
            {
                std::unique_lock<std::mutex> lk(ring.mu);
                ring.cv.wait(lk, [&] { return ring.abort || k - ring.consumed < (u64)SA_IO_SLOTS; });
                if (ring.abort) return;
            }
            const bool ok = full_pread(fd, ring.buf[k % SA_IO_SLOTS], pieces[k].bytes, pieces[k].file_byte);
            std::lock_guard<std::mutex> lk(ring.mu);
            if (!ok) { ring.io_errno = errno ? errno : EIO; ring.cv.notify_all(); return; }
            ring.produced = k + 1;
            ring.cv.notify_all();
        }
    });
    int rc = SA_OK;
    hipError_t he = hipSuccess;
    u64 issued = 0;
    for (u64 k = 0; k < pieces.size() && rc == SA_OK; k++) {
        {
            std::unique_lock<std::mutex> lk(ring.mu);
            ring.cv.wait(lk, [&] { return ring.io_errno || ring.produced > k; });
            if (ring.produced <= k) {
                sa_set_error("reading the index file failed: %s", strerror(ring.io_errno));
                rc = SA_ERR_IO;
                break;
            }
        }
        const int s = (int)(k % SA_IO_SLOTS);
        he = hipMemcpyAsync(d_words + pieces[k].dev_word, ring.buf[s], pieces[k].bytes, hipMemcpyHostToDevice, st);
        if (he == hipSuccess) he = hipEventRecord(ring.done[s], st);
        if (he != hipSuccess) break;
        issued = k + 1;
        // release the oldest slot once the device is done with it, so the reader can refill it
        if (issued >= (u64)SA_IO_SLOTS - 1) {
            const u64 r = issued - ((u64)SA_IO_SLOTS - 1);
            he = hipEventSynchronize(ring.done[r % SA_IO_SLOTS]);
            if (he != hipSuccess) break;
            std::lock_guard<std::mutex> lk(ring.mu);
            ring.consumed = r + 1;
            ring.cv.notify_all();
        }
    }
    {
        std::lock_guard<std::mutex> lk(ring.mu);
        ring.abort = true;
        ring.cv.notify_all();
    }
    io.join();
    const hipError_t hs = hipStreamSynchronize(st);      // slots must be idle before the ring is freed
    if (he == hipSuccess) he = hs;
    if (rc == SA_OK && he != hipSuccess) {
        sa_set_error("%s: copy to the device failed: %s", __FILE__, hipGetErrorString(he));
        rc = SA_ERR_HIP;
    }
    return rc;
}

// device -> file: this thread issues D2H copies, the io thread writes finished slots
int stream_out(int fd, u64 n_words, const u64* d_words, hipStream_t st) {
    Ring ring;
    SA_TRY(ring.init());
    const u64 total = n_words * sizeof(u64);
    const u64 n_pieces = (total + SA_IO_PIECE - 1) / SA_IO_PIECE;
    // produced = pieces whose D2H copy completed (written by this thread), consumed = pieces on disk
    std::thread io([&] {
        for (u64 k = 0; k < n_pieces; k++) {
            {
                std::unique_lock<std::mutex> lk(ring.mu);
                ring.cv.wait(lk, [&] { return ring.abort || ring.produced > k; });
                if (ring.produced <= k) return;
            }
            const u64 off = k * SA_IO_PIECE;
            const u64 bytes = total - off < SA_IO_PIECE ? total - off : SA_IO_PIECE;
            const bool ok = full_pwrite(fd, ring.buf[k % SA_IO_SLOTS], bytes, off);
            std::lock_guard<std::mutex> lk(ring.mu);
            if (!ok) { ring.io_errno = errno ? errno : EIO; ring.cv.notify_all(); return; }
            ring.consumed = k + 1;
            ring.cv.notify_all();
        }
    });
    int rc = SA_OK;
    hipError_t he = hipSuccess;
    for (u64 k = 0; k < n_pieces; k++) {
        {
            std::unique_lock<std::mutex> lk(ring.mu);
            ring.cv.wait(lk, [&] { return ring.io_errno || k - ring.consumed < (u64)SA_IO_SLOTS; });
            if (ring.io_errno) break;
        }
        const u64 off = k * SA_IO_PIECE;
        const u64 bytes = total - off < SA_IO_PIECE ? total - off : SA_IO_PIECE;
        const int s = (int)(k % SA_IO_SLOTS);
        he = hipMemcpyAsync(ring.buf[s], (const char*)d_words + off, bytes, hipMemcpyDeviceToHost, st);
        if (he == hipSuccess) he = hipEventRecord(ring.done[s], st);
        if (he != hipSuccess) break;
        // hand the previous piece to the writer while this one is in flight
        if (k > 0) {
            he = hipEventSynchronize(ring.done[(k - 1) % SA_IO_SLOTS]);
            if (he != hipSuccess) break;
            std::lock_guard<std::mutex> lk(ring.mu);
            ring.produced = k;
            ring.cv.notify_all();
        }
    }
    if (he == hipSuccess) he = hipStreamSynchronize(st);
    {
        std::unique_lock<std::mutex> lk(ring.mu);
        if (he == hipSuccess && !ring.io_errno) {
            ring.produced = n_pieces;
            ring.cv.notify_all();
            ring.cv.wait(lk, [&] { return ring.io_errno || ring.consumed == n_pieces; });
        }
        ring.abort = true;
        ring.cv.notify_all();
    }
    io.join();
    if (ring.io_errno) {
        sa_set_error("writing the index file failed: %s", strerror(ring.io_errno));
        rc = SA_ERR_IO;
    } else if (he != hipSuccess) {
        sa_set_error("%s: copy from the device failed: %s", __FILE__, hipGetErrorString(he));
        rc = SA_ERR_HIP;
    }
    return rc;
}

}  // namespace

extern "C" int sa_index_create_from_file(int device, uint64_t n_docs, uint64_t doc_base, uint32_t n_terms,
                                         const char* path, const uint64_t* term_src_off, const uint64_t* term_len,
                                         const float* doc_lens, float avg_doc_len, uint64_t corpus_size,
                                         uint32_t tile_docs, sa_index_t** out) {
    SA_ARG(out, "out is null");
    SA_ARG(path, "path is null");
    SA_ARG(n_terms == 0 || (term_src_off && term_len), "term_src_off / term_len is null");
    SA_ARG(n_docs == 0 || doc_lens, "doc_lens is null");
    SA_ARG(n_docs <= (1ull << 28), "a shard holds at most 2^28 docs (28-bit roaringish key)");
    if (tile_docs == 0) tile_docs = 8192;
    SA_ARG(tile_docs == 1024 || tile_docs == 2048 || tile_docs == 4096 || tile_docs == 8192 ||
               tile_docs == 16384 || tile_docs == 32768,
           "tile_docs must be 1024, 2048, 4096, 8192, 16384 or 32768");

    const int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) { sa_set_error("cannot open %s: %s", path, strerror(errno)); return SA_ERR_IO; }
    struct Closer { int fd; ~Closer() { close(fd); } } closer{fd};
    struct stat sb;
    if (fstat(fd, &sb) != 0) { sa_set_error("cannot stat %s: %s", path, strerror(errno)); return SA_ERR_IO; }
    const u64 file_words = (u64)sb.st_size / sizeof(u64);

    // device layout: terms back to back in id order, whatever their order in the file
    std::vector<u64> term_off((size_t)n_terms + 1, 0);
    std::vector<Piece> pieces;
    u64 W = 0;
    for (u32 t = 0; t < n_terms; t++) {
        const u64 len = term_len[t], src = term_src_off[t];
        if (len && (src > file_words || len > file_words - src)) {
            sa_set_error("term %u (offset %llu, length %llu words) lies outside %s (%llu words)", t,
                         (unsigned long long)src, (unsigned long long)len, path, (unsigned long long)file_words);
            return SA_ERR_ARG;
        }
        term_off[t] = W;
        u64 done = 0;
        while (done < len) {
            // grow the last piece when this term continues it in the file, else open a new one
            if (!pieces.empty()) {
                Piece& b = pieces.back();
                if (b.file_byte + b.bytes == (src + done) * sizeof(u64) && b.bytes < SA_IO_PIECE) {
                    const u64 room = (SA_IO_PIECE - b.bytes) / sizeof(u64);
                    const u64 n = len - done < room ? len - done : room;
                    b.bytes += n * sizeof(u64);
                    done += n;
                    continue;
                }
            }
            const u64 cap = SA_IO_PIECE / sizeof(u64);
            const u64 n = len - done < cap ? len - done : cap;
            pieces.push_back({(src + done) * sizeof(u64), W + done, n * sizeof(u64)});
            done += n;
        }
        W += len;
    }
    term_off[n_terms] = W;

    sa_index* ix = new (std::nothrow) sa_index();
    if (!ix) { sa_set_error("out of host memory"); return SA_ERR_NOMEM; }
    ix->device = device;
    ix->n_docs = n_docs; ix->doc_base = doc_base; ix->corpus_size = corpus_size;
    ix->n_terms = n_terms; ix->avg_doc_len = avg_doc_len; ix->n_words = W;
    ix->tile_docs = tile_docs;
    ix->h_term_off = term_off;
    auto build = [&]() -> int {
        SA_TRY(sa_index_setup(ix, doc_lens));
        SA_HIP(hipMalloc(&ix->d_words, (W ? W : 1) * sizeof(u64)));
        SA_HIP(hipMalloc(&ix->d_term_off, ((size_t)n_terms + 1) * sizeof(u64)));
        SA_HIP(hipMemcpyAsync(ix->d_term_off, ix->h_term_off.data(), ((size_t)n_terms + 1) * sizeof(u64),
                              hipMemcpyHostToDevice, ix->stream));
        SA_TRY(stream_in(fd, pieces, ix->d_words, ix->stream));
        return sa_index_derive(ix);
    };
    const int rc = build();
    if (rc != SA_OK) { sa_index_free(ix); return rc; }
    *out = ix;
    return SA_OK;
}

extern "C" int sa_index_save(sa_index_t* ix, const char* path) {
    SA_ARG(ix && path, "null argument");
    std::lock_guard<std::mutex> g(ix->mu);
    SA_HIP(hipSetDevice(ix->device));
    const int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0644);
    if (fd < 0) { sa_set_error("cannot create %s: %s", path, strerror(errno)); return SA_ERR_IO; }
    int rc = stream_out(fd, ix->n_words, ix->d_words, ix->stream);
    if (close(fd) != 0 && rc == SA_OK) { sa_set_error("closing %s failed: %s", path, strerror(errno)); rc = SA_ERR_IO; }
    return rc;
}
