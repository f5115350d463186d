// tests/hipemu/hip/hip_runtime.h -- TEST INFRASTRUCTURE ONLY.
//
// A tiny host-side stand-in for the subset of the HIP runtime + device language that
// searcharray_amd/csrc/*.hip uses, so the UNMODIFIED kernel sources can be compiled with
// g++ (-I tests/hipemu -x c++) and their logic exercised on a machine without a GPU
// (tests/test_emu_*.py).  The product never sees this header: the real build uses hipcc
// and /opt/rocm's <hip/hip_runtime.h>, and searcharray_amd/_lib.py only ever loads the
// gfx950 library.
//
// Execution model: blocks run one after another; the threads of a block are cooperative
// fibers (ucontext) scheduled round-robin by one OS thread.  A fiber runs until it reaches
// __syncthreads() or a wave-collective (__shfl*, __ballot, __any, __all, readfirstlane),
// which makes the schedule deterministic and adversarial (thread 0 runs a whole phase
// before thread 1 starts), so missing barriers show up as wrong results.  Wave width is 64.
// A wave collective completes when every live lane of the wave is blocked; the lanes
// blocked on the collective are its participants (models the exec mask under divergence).
#pragma once
#include <ucontext.h>
#include <mutex>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <chrono>
#include <functional>
#include <vector>

#define __global__
#define SA_AS_GLOBAL                                 // (address spaces are the device compiler's business)
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)
#define __restrict__ __restrict

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3_emu { unsigned x, y, z; };
struct alignas(16) float4 { float x, y, z, w; };
inline float4 make_float4(float x, float y, float z, float w) { float4 v; v.x = x; v.y = y; v.z = z; v.w = w; return v; }
struct alignas(16) uint4 { unsigned x, y, z, w; };
inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { uint4 v; v.x = x; v.y = y; v.z = z; v.w = w; return v; }

typedef int hipError_t;
typedef struct hipemu_stream* hipStream_t;
typedef struct hipemu_event* hipEvent_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorInvalidConfiguration = 9 };
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost,
                     hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum { hipStreamNonBlocking = 1, hipEventDefault = 0 };

struct hipDeviceProp_t {
    char name[256]; char gcnArchName[256]; int multiProcessorCount; size_t totalGlobalMem;
};

namespace hipemu {

constexpr int WAVE = 64;
enum State { RUNNABLE, WAIT_BLOCK, WAIT_WAVE, DONE };
enum WaveOp { OP_NONE, OP_SHFL, OP_BALLOT, OP_FIRST, OP_DPP };

struct Fiber {
    ucontext_t ctx;
    char* stack = nullptr;
    State state = DONE;
    uint3_emu tid{0, 0, 0};
    int linear = 0;
    // wave-collective mailbox
    WaveOp op = OP_NONE;
    uint64_t val = 0;      // posted value (shfl payload / predicate)
    int src = 0;           // shfl source lane
    uint64_t old = 0;      // DPP: value kept when the lane is masked off / source invalid
    int ctrl = 0, row_mask = 0xf, bank_mask = 0xf, bound_ctrl = 0;
    uint64_t result = 0;
};

struct Machine {
    std::vector<Fiber> fibers;
    ucontext_t sched;
    Fiber* cur = nullptr;
    dim3 gridDim, blockDim, blockIdx;
    std::function<void()> body;
    size_t stack_size = 128 * 1024;
};

inline Machine& M() { static Machine m; return m; }

inline void yield_to_sched() {
    Machine& m = M();
    swapcontext(&m.cur->ctx, &m.sched);
}

inline void fiber_entry() {
    Machine& m = M();
    m.body();
    m.cur->state = DONE;
    swapcontext(&m.cur->ctx, &m.sched);
}

inline void die(const char* msg) {
    fprintf(stderr, "[hipemu] FATAL: %s\n", msg);
    abort();
}

inline void run_block() {
    Machine& m = M();
    const int nthreads = (int)(m.blockDim.x * m.blockDim.y * m.blockDim.z);
    if ((int)m.fibers.size() < nthreads) m.fibers.resize(nthreads);
    for (int i = 0; i < nthreads; i++) {
        Fiber& f = m.fibers[i];
        if (!f.stack) f.stack = (char*)malloc(m.stack_size);
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack;
        f.ctx.uc_stack.ss_size = m.stack_size;
        f.ctx.uc_link = &m.sched;
        makecontext(&f.ctx, (void (*)())fiber_entry, 0);
        f.state = RUNNABLE;
        f.op = OP_NONE;
        f.linear = i;
        f.tid.x = i % m.blockDim.x;
        f.tid.y = (i / m.blockDim.x) % m.blockDim.y;
        f.tid.z = i / (m.blockDim.x * m.blockDim.y);
    }
    int alive = nthreads;
    while (alive > 0) {
        bool progressed = false;
        for (int i = 0; i < nthreads; i++) {
            Fiber& f = m.fibers[i];
            if (f.state != RUNNABLE) continue;
            m.cur = &f;
            swapcontext(&m.sched, &f.ctx);
            progressed = true;
            if (f.state == DONE) alive--;
        }
        if (alive == 0) break;
        // every live fiber is blocked now: resolve wave collectives
        bool resolved = false;
        const int nwaves = (nthreads + WAVE - 1) / WAVE;
        for (int w = 0; w < nwaves; w++) {
            int lo = w * WAVE, hi = lo + WAVE < nthreads ? lo + WAVE : nthreads;
            WaveOp op = OP_NONE;
            for (int i = lo; i < hi; i++) {
                Fiber& f = m.fibers[i];
                if (f.state == WAIT_WAVE) {
                    if (op == OP_NONE) op = f.op;
                    else if (op != f.op) die("lanes of one wave blocked on different collectives");
                }
            }
            if (op == OP_NONE) continue;
            uint64_t ballot = 0;
            int first = -1;
            for (int i = lo; i < hi; i++) {
                Fiber& f = m.fibers[i];
                if (f.state != WAIT_WAVE) continue;
                if (first < 0) first = i;
                if (f.val) ballot |= (1ull << (i - lo));
            }
            for (int i = lo; i < hi; i++) {
                Fiber& f = m.fibers[i];
                if (f.state != WAIT_WAVE) continue;
                if (op == OP_BALLOT) f.result = ballot;
                else if (op == OP_FIRST) f.result = m.fibers[first].val;
                else if (op == OP_DPP) {
                    // gfx9 DPP semantics for the controls the kernels use
                    const int l = i - lo, row = l >> 4, bank = (l & 15) >> 2;
                    int s = -1;
                    if (f.ctrl >= 0x111 && f.ctrl <= 0x11F) { const int n = f.ctrl - 0x110; if ((l & 15) >= n) s = l - n; }
                    else if (f.ctrl == 0x142) { if (row >= 1) s = (row - 1) * 16 + 15; }
                    else if (f.ctrl == 0x143) { if (row >= 2) s = 31; }
                    else die("unsupported dpp_ctrl in emulator");
                    const bool enabled = ((f.row_mask >> row) & 1) && ((f.bank_mask >> bank) & 1);
                    const bool valid = s >= 0 && lo + s < hi && m.fibers[lo + s].state == WAIT_WAVE;
                    if (!enabled) f.result = f.old;
                    else if (valid) f.result = m.fibers[lo + s].val;
                    else f.result = f.bound_ctrl ? 0 : f.old;
                }
                else {  // OP_SHFL
                    int s = lo + (f.src & (WAVE - 1));
                    if (s < hi && m.fibers[s].state == WAIT_WAVE) f.result = m.fibers[s].val;
                    else f.result = f.val;   // reading an inactive lane: keep own value
                }
            }
            for (int i = lo; i < hi; i++) {
                Fiber& f = m.fibers[i];
                if (f.state == WAIT_WAVE) { f.state = RUNNABLE; f.op = OP_NONE; }
            }
            resolved = true;
        }
        if (resolved) continue;
        // no wave work: every live fiber must be at the block barrier
        bool all_block = true;
        for (int i = 0; i < nthreads; i++)
            if (m.fibers[i].state != WAIT_BLOCK && m.fibers[i].state != DONE) all_block = false;
        if (all_block) {
            for (int i = 0; i < nthreads; i++)
                if (m.fibers[i].state == WAIT_BLOCK) m.fibers[i].state = RUNNABLE;
            continue;
        }
        if (!progressed) die("deadlock: no runnable fiber");
    }
}

// one kernel at a time: the machine (fibers, __shared__ statics) is process-global, and the in-process
// communicator of the test build drives one shard per host thread
inline std::mutex& launch_mutex() { static std::mutex mu; return mu; }

// like the runtime: a launch with an empty grid or block does not run and leaves an error for hipGetLastError()
inline int& last_launch_error() { static thread_local int e = 0; return e; }
inline void launch(dim3 grid, dim3 block, std::function<void()> body) {
    if (grid.x == 0 || grid.y == 0 || grid.z == 0 || block.x * block.y * block.z == 0 || block.x * block.y * block.z > 1024) {
        last_launch_error() = 9;
        return;
    }
    std::lock_guard<std::mutex> launch_guard(launch_mutex());
    Machine& m = M();
    m.gridDim = grid;
    m.blockDim = block;
    m.body = std::move(body);
    for (unsigned z = 0; z < grid.z; z++)
        for (unsigned y = 0; y < grid.y; y++)
            for (unsigned x = 0; x < grid.x; x++) {
                m.blockIdx = dim3(x, y, z);
                run_block();
            }
}

inline uint64_t wave_collective(WaveOp op, uint64_t val, int src) {
    Machine& m = M();
    Fiber* f = m.cur;
    f->op = op; f->val = val; f->src = src; f->state = WAIT_WAVE;
    yield_to_sched();
    return f->result;
}

template <class T> inline uint64_t to_bits(T v) { uint64_t b = 0; memcpy(&b, &v, sizeof(T)); return b; }
template <class T> inline T from_bits(uint64_t b) { T v; memcpy(&v, &b, sizeof(T)); return v; }

}  // namespace hipemu

#define threadIdx (hipemu::M().cur->tid)
#define blockIdx (hipemu::M().blockIdx)
#define blockDim (hipemu::M().blockDim)
#define gridDim (hipemu::M().gridDim)
#define warpSize 64

inline void __syncthreads() {
    hipemu::M().cur->state = hipemu::WAIT_BLOCK;
    hipemu::yield_to_sched();
}
inline void __threadfence() {}
inline void __threadfence_block() {}

inline int __lane_id() { return hipemu::M().cur->linear & 63; }

template <class T> inline T __shfl(T v, int src, int width = 64) {
    int lane = __lane_id();
    int base = lane & ~(width - 1);
    return hipemu::from_bits<T>(hipemu::wave_collective(hipemu::OP_SHFL, hipemu::to_bits(v), base + (src & (width - 1))));
}
template <class T> inline T __shfl_xor(T v, int mask, int width = 64) {
    int lane = __lane_id();
    int s = lane ^ mask;
    if ((s & ~(width - 1)) != (lane & ~(width - 1))) s = lane;
    return hipemu::from_bits<T>(hipemu::wave_collective(hipemu::OP_SHFL, hipemu::to_bits(v), s));
}
template <class T> inline T __shfl_down(T v, unsigned d, int width = 64) {
    int lane = __lane_id();
    int s = lane + (int)d;
    if ((s & ~(width - 1)) != (lane & ~(width - 1))) s = lane;
    return hipemu::from_bits<T>(hipemu::wave_collective(hipemu::OP_SHFL, hipemu::to_bits(v), s));
}
template <class T> inline T __shfl_up(T v, unsigned d, int width = 64) {
    int lane = __lane_id();
    int s = lane - (int)d;
    if (s < 0 || (s & ~(width - 1)) != (lane & ~(width - 1))) s = lane;
    return hipemu::from_bits<T>(hipemu::wave_collective(hipemu::OP_SHFL, hipemu::to_bits(v), s));
}
inline unsigned long long __ballot(int pred) {
    return hipemu::wave_collective(hipemu::OP_BALLOT, pred ? 1 : 0, 0);
}
inline unsigned long long __builtin_amdgcn_ballot_w64(bool pred) { return __ballot(pred ? 1 : 0); }
inline int __any(int pred) { return __ballot(pred) != 0; }
inline int __all(int pred) {
    // all participating lanes true <=> no participating lane false
    return __ballot(!pred) == 0;
}
inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
    hipemu::Fiber* f = hipemu::M().cur;
    f->old = (uint64_t)(uint32_t)old; f->ctrl = ctrl; f->row_mask = row_mask; f->bank_mask = bank_mask;
    f->bound_ctrl = bound_ctrl ? 1 : 0;
    return (int)(uint32_t)hipemu::wave_collective(hipemu::OP_DPP, (uint64_t)(uint32_t)src, 0);
}
// scheduling barrier on the GPU; here the lanes of a wave are fibers, so it must really line them up
inline void __builtin_amdgcn_wave_barrier() { (void)__ballot(1); }
inline void __builtin_amdgcn_s_waitcnt(int) {}          // memory is synchronous here
inline void __builtin_amdgcn_sched_barrier(int) {}      // instruction scheduling only
inline int __builtin_amdgcn_readlane(int v, int lane) {
    return (int)(uint32_t)hipemu::wave_collective(hipemu::OP_SHFL, (uint64_t)(uint32_t)v, lane);
}
// ds_bpermute_b32: lane L reads the value of lane (addr / 4) mod 64
inline int __builtin_amdgcn_ds_bpermute(int addr, int v) {
    return (int)(uint32_t)hipemu::wave_collective(hipemu::OP_SHFL, (uint64_t)(uint32_t)v, (addr >> 2) & 63);
}
inline int __builtin_amdgcn_readfirstlane(int v) {
    return (int)hipemu::wave_collective(hipemu::OP_FIRST, (uint64_t)(uint32_t)v, 0);
}

// v_mbcnt_lo / v_mbcnt_hi: base + set bits of the mask half below this lane
inline unsigned __builtin_amdgcn_mbcnt_lo(unsigned mask, unsigned base) {
    const int lane = __lane_id();
    return base + (unsigned)__builtin_popcount(lane >= 32 ? mask : (mask & ((1u << lane) - 1u)));
}
inline unsigned __builtin_amdgcn_mbcnt_hi(unsigned mask, unsigned base) {
    const int lane = __lane_id();
    return base + (lane > 32 ? (unsigned)__builtin_popcount(mask & ((1u << (lane - 32)) - 1u)) : 0u);
}

// buffer resources (raw buffer loads: base + 32-bit byte offset, a read past the end returns 0)
struct __amdgpu_buffer_rsrc_t { const char* base; unsigned bytes; };
inline __amdgpu_buffer_rsrc_t __builtin_amdgcn_make_buffer_rsrc(void* p, short, int num_bytes, int) {
    __amdgpu_buffer_rsrc_t r; r.base = (const char*)p; r.bytes = (unsigned)num_bytes; return r;
}
typedef unsigned int hipemu_v2u __attribute__((vector_size(8)));
typedef unsigned int hipemu_v4u __attribute__((vector_size(16)));
inline unsigned __builtin_amdgcn_raw_buffer_load_b32(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, int) {
    unsigned v = 0; if ((unsigned long long)voff + soff + 4 <= r.bytes) memcpy(&v, r.base + voff + soff, 4); return v;
}
inline hipemu_v2u __builtin_amdgcn_raw_buffer_load_b64(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, int) {
    hipemu_v2u v = {0, 0}; if ((unsigned long long)voff + soff + 8 <= r.bytes) memcpy(&v, r.base + voff + soff, 8); return v;
}
inline hipemu_v4u __builtin_amdgcn_raw_buffer_load_b128(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, int) {
    hipemu_v4u v = {0, 0, 0, 0}; if ((unsigned long long)voff + soff + 16 <= r.bytes) memcpy(&v, r.base + voff + soff, 16); return v;
}

inline int __popc(unsigned x) { return __builtin_popcount(x); }
inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
inline int __ffs(int x) { return __builtin_ffs(x); }
inline int __ffsll(long long x) { return __builtin_ffsll(x); }
inline int __clz(int x) { return x ? __builtin_clz((unsigned)x) : 32; }
inline int __clzll(long long x) { return x ? __builtin_clzll((unsigned long long)x) : 64; }
inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
inline int __float_as_int(float f) { int u; memcpy(&u, &f, 4); return u; }
inline float __int_as_float(int u) { float f; memcpy(&f, &u, 4); return f; }
// v_cvt_u32_f32: truncates toward zero, saturates, NaN -> 0
inline unsigned __float2uint_rz(float x) { return x >= 4294967296.f ? 0xFFFFFFFFu : (x > 0.f ? (unsigned)x : 0u); }
inline float __fdiv_rn(float a, float b) { return a / b; }
inline float __fsqrt_rn(float a) { return sqrtf(a); }
inline float __fmul_rn(float a, float b) { return a * b; }
inline float __fadd_rn(float a, float b) { return a + b; }
inline float __fsub_rn(float a, float b) { return a - b; }
inline double __dadd_rn(double a, double b) { return a + b; }
inline double __dsub_rn(double a, double b) { return a - b; }
inline double __dmul_rn(double a, double b) { return a * b; }
template <class T> inline T __ldg(const T* p) { return *p; }
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __hip_atomic_load(ptr, order, scope) (*(ptr))

// atomics (single OS thread: plain read-modify-write)
template <class T> inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
template <class T> inline T atomicSub(T* p, T v) { T o = *p; *p = o - v; return o; }
template <class T> inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <class T> inline T atomicMin(T* p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <class T> inline T atomicOr(T* p, T v) { T o = *p; *p = o | v; return o; }
template <class T> inline T atomicAnd(T* p, T v) { T o = *p; *p = o & v; return o; }
inline float unsafeAtomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }
template <class T> inline T atomicExch(T* p, T v) { T o = *p; *p = v; return o; }
template <class T> inline T atomicCAS(T* p, T cmp, T v) { T o = *p; if (o == cmp) *p = v; return o; }

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    hipemu::launch(dim3(grid), dim3(block), [=]() { kernel(__VA_ARGS__); })

// ---- host runtime subset ----
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
    memset(p, 0, sizeof(*p)); strcpy(p->name, "hipemu"); strcpy(p->gcnArchName, "host");
    p->multiProcessorCount = 4; p->totalGlobalMem = (size_t)8 << 30; return hipSuccess;
}
inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
template <class T> inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }
inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { return hipMalloc(p, n); }
template <class T> inline hipError_t hipHostMalloc(T** p, size_t n, unsigned = 0) { return hipMalloc((void**)p, n); }
inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { if (n) memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t = 0) { if (n) memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemset(void* d, int v, size_t n) { if (n) memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t = 0) { if (n) memset(d, v, n); return hipSuccess; }
inline hipError_t hipStreamCreate(hipStream_t* s) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipGetLastError() { const int e = hipemu::last_launch_error(); hipemu::last_launch_error() = 0; return (hipError_t)e; }
inline hipError_t hipPeekAtLastError() { return (hipError_t)hipemu::last_launch_error(); }
inline const char* hipGetErrorString(hipError_t e) { return (int)e == 9 ? "invalid configuration argument" : "hipemu error"; }
struct hipemu_event { std::chrono::steady_clock::time_point t; };
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new hipemu_event(); return hipSuccess; }
enum { hipEventDisableTiming = 2 };
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new hipemu_event(); return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = 0) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned = 0) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count(); return hipSuccess;
}
template <class K> inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, K, int, size_t) { *n = 2; return hipSuccess; }
inline hipError_t hipMemGetInfo(size_t* f, size_t* t) { *f = (size_t)4 << 30; *t = (size_t)8 << 30; return hipSuccess; }
