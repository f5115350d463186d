// On-disk index <-> HBM.
//
// The reference persists an index as ONE raw file of uint64 roaringish words -- ArrayDict.data.tofile
// (searcharray/phrase/memmap_arrays.py:158-161) -- plus a metadata dict {term id: {offset, length}}
// in element units (memmap_arrays.py:28-54) that travels in the pickle; reading it back is a numpy
// memmap whose pages fault in term by term as queries touch them (memmap_arrays.py:163-165).
//
// Here the file is streamed straight into HBM: file threads pread 16 MiB pieces into a ring of
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

// the options of the handle a file routine works for (sa_index_save: the index's), over what the calling thread's defaults say (the
// loading routines run before an index handle exists)
static thread_local const sa_options_t* t_io_handle_opts = nullptr;
static sa_options_t sa_io_opts() { return sa_options_for_new_handle(t_io_handle_opts); }
struct SaIoHandleOpts { SaIoHandleOpts(const sa_options_t* o) { t_io_handle_opts = o; } ~SaIoHandleOpts() { t_io_handle_opts = nullptr; } };
// bytes per staged piece (option io_piece_bytes: test hook, lets small files exercise the ring's wrap-around)
static size_t sa_io_piece() {
    size_t v = 4u << 20;      // larger pieces only add page-locking time (~0.5 ms per MiB of ring), measured
    const sa_options_t o = sa_io_opts();
    if (sa_opt_is_set(o.io_piece_bytes)) {
        const size_t x = (size_t)o.io_piece_bytes;
        if (x >= 64 && x <= (256u << 20)) v = x & ~(size_t)7;
    }
    return v;
}
constexpr int SA_IO_SLOTS = 12;
// file threads: a single pread / pwrite stream tops out near 10 GB/s (option io_threads: 1..SA_IO_SLOTS / 2)
static int sa_io_workers() {
    const int v = (int)sa_opt(sa_io_opts().io_threads, 4);
    return v < 1 ? 1 : (v > SA_IO_SLOTS / 2 ? SA_IO_SLOTS / 2 : v);
}

struct Piece { u64 file_byte; u64 dev_word; u64 bytes; };

// Ring of pinned staging buffers shared by the file threads (piece k belongs to thread k % WORKERS and
// lives in slot k % SLOTS) and the thread that drives the copy engine.
struct Ring {
    void* buf[SA_IO_SLOTS] = {};
    hipEvent_t done[SA_IO_SLOTS] = {};      // device finished with the slot
    std::mutex mu;
    std::condition_variable cv;
    u64 staged[SA_IO_SLOTS] = {};           // piece index + 1 whose bytes the slot holds (file side done / copy done)
    u64 released = 0;                       // pieces whose slot may be reused
    int io_errno = 0;
    bool abort = false;

    size_t piece = 0;

    // Page-locking the ring costs more than streaming a small index through it, so one set of buffers is
    // parked here between calls (per process; a second concurrent stream allocates its own).
    struct Parked { std::mutex mu; void* buf[SA_IO_SLOTS] = {}; size_t piece = 0; bool full = false; };
    static Parked& parked() { static Parked p; return p; }

    int init() {
        piece = sa_io_piece();
        {
            Parked& pk = parked();
            std::lock_guard<std::mutex> lk(pk.mu);
            if (pk.full && pk.piece == piece) {
                for (int i = 0; i < SA_IO_SLOTS; i++) { buf[i] = pk.buf[i]; pk.buf[i] = nullptr; }
                pk.full = false;
            }
        }
        for (int i = 0; i < SA_IO_SLOTS; i++) {
            if (!buf[i]) SA_HIP(hipHostMalloc(&buf[i], piece));
            SA_HIP(hipEventCreateWithFlags(&done[i], hipEventDisableTiming));
        }
        return SA_OK;
    }
    ~Ring() {
        for (int i = 0; i < SA_IO_SLOTS; i++)
            if (done[i]) hipEventDestroy(done[i]);
        bool complete = true;
        for (int i = 0; i < SA_IO_SLOTS; i++) complete = complete && buf[i];
        if (complete) {
            Parked& pk = parked();
            std::lock_guard<std::mutex> lk(pk.mu);
            if (pk.full && pk.piece != piece) {                  // a ring of another piece size: replace it
                for (int i = 0; i < SA_IO_SLOTS; i++) { hipHostFree(pk.buf[i]); pk.buf[i] = nullptr; }
                pk.full = false;
            }
            if (!pk.full) {
                for (int i = 0; i < SA_IO_SLOTS; i++) { pk.buf[i] = buf[i]; buf[i] = nullptr; }
                pk.piece = piece;
                pk.full = true;
            }
        }
        for (int i = 0; i < SA_IO_SLOTS; i++)
            if (buf[i]) hipHostFree(buf[i]);
    }
    void stop(std::vector<std::thread>& workers) {
        {
            std::lock_guard<std::mutex> lk(mu);
            abort = true;
            cv.notify_all();
        }
        for (std::thread& t : workers) t.join();
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

// file -> device: the file threads fill slots, this thread issues the copies in piece order
int stream_in(int fd, const std::vector<Piece>& pieces, u64* d_words, hipStream_t st) {
    Ring ring;
    SA_TRY(ring.init());
    const u64 n = pieces.size();
    const int SA_IO_WORKERS = sa_io_workers();
    std::vector<std::thread> workers;
    for (int w = 0; w < SA_IO_WORKERS; w++) {
        workers.emplace_back([&, w] {
            for (u64 k = (u64)w; k < n; k += SA_IO_WORKERS) {
                {
                    std::unique_lock<std::mutex> lk(ring.mu);
                    ring.cv.wait(lk, [&] { return ring.abort || ring.io_errno || k < ring.released + SA_IO_SLOTS; });
                    if (ring.abort || ring.io_errno) return;
                }
                const bool ok = full_pread(fd, ring.buf[k % SA_IO_SLOTS], pieces[k].bytes, pieces[k].file_byte);
                std::lock_guard<std::mutex> lk(ring.mu);
                if (!ok) { if (!ring.io_errno) ring.io_errno = errno ? errno : EIO; }
                else ring.staged[k % SA_IO_SLOTS] = k + 1;
                ring.cv.notify_all();
                if (!ok) return;
            }
        });
    }
    int rc = SA_OK;
    hipError_t he = hipSuccess;
    const u64 lag = SA_IO_SLOTS / 2;                          // copies kept in flight before a slot is released
    for (u64 k = 0; k < n; k++) {
        const int s = (int)(k % SA_IO_SLOTS);
        {
            std::unique_lock<std::mutex> lk(ring.mu);
            ring.cv.wait(lk, [&] { return ring.io_errno || ring.staged[s] == k + 1; });
            if (ring.staged[s] != k + 1) {
                sa_set_error("reading the index file failed: %s", strerror(ring.io_errno));
                rc = SA_ERR_IO;
                break;
            }
        }
        he = hipMemcpyAsync(d_words + pieces[k].dev_word, ring.buf[s], pieces[k].bytes, hipMemcpyHostToDevice, st);
        if (he == hipSuccess) he = hipEventRecord(ring.done[s], st);
        if (he != hipSuccess) break;
        if (k + 1 >= lag) {                                   // hand the oldest in-flight slot back to the file threads
            const u64 r = k + 1 - lag;
            he = hipEventSynchronize(ring.done[r % SA_IO_SLOTS]);
            if (he != hipSuccess) break;
            std::lock_guard<std::mutex> lk(ring.mu);
            ring.released = r + 1;
            ring.cv.notify_all();
        }
    }
    ring.stop(workers);
    const hipError_t hs = hipStreamSynchronize(st);           // slots must be idle before the ring is freed
    if (he == hipSuccess) he = hs;
    if (rc == SA_OK && he != hipSuccess) {
        sa_set_error("%s: copy to the device failed: %s", __FILE__, hipGetErrorString(he));
        rc = SA_ERR_HIP;
    }
    return rc;
}

// device -> file: this thread issues D2H copies in piece order, the file threads write finished slots
int stream_out(int fd, u64 n_words, const u64* d_words, hipStream_t st) {
    Ring ring;
    SA_TRY(ring.init());
    const u64 total = n_words * sizeof(u64);
    const u64 SA_IO_PIECE = ring.piece;
    const u64 n = (total + SA_IO_PIECE - 1) / SA_IO_PIECE;
    const int SA_IO_WORKERS = sa_io_workers();
    u64 written = 0;                                          // pieces on disk (any order), guarded by ring.mu
    u64 on_disk[SA_IO_SLOTS] = {};                            // piece index + 1 last written from the slot
    std::vector<std::thread> workers;
    for (int w = 0; w < SA_IO_WORKERS; w++) {
        workers.emplace_back([&, w] {
            for (u64 k = (u64)w; k < n; k += SA_IO_WORKERS) {
                const int s = (int)(k % SA_IO_SLOTS);
                {
                    std::unique_lock<std::mutex> lk(ring.mu);
                    ring.cv.wait(lk, [&] { return ring.abort || ring.io_errno || ring.staged[s] == k + 1; });
                    if (ring.staged[s] != k + 1) return;
                }
                const u64 off = k * SA_IO_PIECE;
                const u64 bytes = total - off < SA_IO_PIECE ? total - off : SA_IO_PIECE;
                const bool ok = full_pwrite(fd, ring.buf[s], bytes, off);
                std::lock_guard<std::mutex> lk(ring.mu);
                if (!ok) { if (!ring.io_errno) ring.io_errno = errno ? errno : EIO; }
                else { on_disk[s] = k + 1; written++; }
                ring.cv.notify_all();
                if (!ok) return;
            }
        });
    }
    hipError_t he = hipSuccess;
    auto publish = [&](u64 k) -> hipError_t {                 // piece k's copy has landed: hand it to its writer
        const hipError_t e = hipEventSynchronize(ring.done[k % SA_IO_SLOTS]);
        if (e != hipSuccess) return e;
        std::lock_guard<std::mutex> lk(ring.mu);
        ring.staged[k % SA_IO_SLOTS] = k + 1;
        ring.cv.notify_all();
        return hipSuccess;
    };
    u64 issued = 0, published = 0;
    for (u64 k = 0; k < n && he == hipSuccess; k++) {
        const int s = (int)(k % SA_IO_SLOTS);
        // the slot's previous piece (k - SLOTS) must be on disk; publish copies that landed while we wait
        while (k >= SA_IO_SLOTS && he == hipSuccess) {
            {
                std::lock_guard<std::mutex> lk(ring.mu);
                if (ring.io_errno || on_disk[s] == k - SA_IO_SLOTS + 1) break;
            }
            if (published < issued) { he = publish(published); published++; continue; }
            std::unique_lock<std::mutex> lk(ring.mu);
            ring.cv.wait(lk, [&] { return ring.io_errno || on_disk[s] == k - SA_IO_SLOTS + 1; });
        }
        {
            std::lock_guard<std::mutex> lk(ring.mu);
            if (ring.io_errno) break;
        }
        if (he != hipSuccess) break;
        const u64 off = k * SA_IO_PIECE;
        const u64 bytes = total - off < SA_IO_PIECE ? total - off : SA_IO_PIECE;
        he = hipMemcpyAsync(ring.buf[s], (const char*)d_words + off, bytes, hipMemcpyDeviceToHost, st);
        if (he == hipSuccess) he = hipEventRecord(ring.done[s], st);
        if (he != hipSuccess) break;
        issued = k + 1;
        while (he == hipSuccess && issued - published > 2) { he = publish(published); published++; }
    }
    while (he == hipSuccess && published < issued) { he = publish(published); published++; }
    {
        std::unique_lock<std::mutex> lk(ring.mu);
        if (he == hipSuccess && issued == n)
            ring.cv.wait(lk, [&] { return ring.io_errno || written == n; });
    }
    ring.stop(workers);
    const hipError_t hs = hipStreamSynchronize(st);
    if (he == hipSuccess) he = hs;
    int rc = SA_OK;
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
    if (tile_docs == 0) tile_docs = SA_DEFAULT_TILE_DOCS;
    SA_ARG(tile_docs == 1024 || tile_docs == 2048 || tile_docs == 4096 || tile_docs == 8192 ||
               tile_docs == 16384 || tile_docs == 32768,
           "tile_docs must be 1024, 2048, 4096, 8192, 16384 or 32768");

    const int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) { sa_set_error("cannot open %s: %s", path, strerror(errno)); return SA_ERR_IO; }
    struct Closer { int fd; ~Closer() { close(fd); } } closer{fd};
    struct stat sb;
    if (fstat(fd, &sb) != 0) { sa_set_error("cannot stat %s: %s", path, strerror(errno)); return SA_ERR_IO; }
    const u64 file_words = (u64)sb.st_size / sizeof(u64);

    const u64 SA_IO_PIECE = sa_io_piece();
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
    ix->opts = sa_options_for_new_handle(nullptr);
    ix->device = device;
    ix->n_docs = n_docs; ix->doc_base = doc_base; ix->corpus_size = corpus_size;
    ix->n_terms = n_terms; ix->avg_doc_len = avg_doc_len; ix->n_words = W;
    ix->tile_docs = tile_docs;
    ix->h_term_off = term_off;
    auto build = [&]() -> int {
        SA_TRY(sa_index_setup(ix, doc_lens));
        SA_HIP(hipMalloc(&ix->d_words, (W + SA_WORDS_PAD) * sizeof(u64)));
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
    SaIoHandleOpts io_opts(&ix->opts);                         // (io_piece_bytes / io_threads of the INDEX apply to its save)
    SA_HIP(hipSetDevice(ix->device));
    const int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0644);
    if (fd < 0) { sa_set_error("cannot create %s: %s", path, strerror(errno)); return SA_ERR_IO; }
    int rc = stream_out(fd, ix->n_words, ix->d_words, ix->stream);
    if (close(fd) != 0 && rc == SA_OK) { sa_set_error("closing %s failed: %s", path, strerror(errno)); rc = SA_ERR_IO; }
    return rc;
}
