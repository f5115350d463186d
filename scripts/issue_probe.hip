// scripts/issue_probe.hip -- MEASUREMENT TOOL (not part of the library).
//
// What bounds a kernel whose waves interleave scalar and vector instructions 1 : 1 ("Next" 1a)?
// Resident waves of ONE SIMD run pure-VALU, pure-SALU, mixed and LDS read-modify-write loops; the time per
// instruction says whether VALU of one wave and SALU of another share an issue slot (sum) or not (max), what a
// single wave can issue, and what the dependent LDS round trip of the BM25 overlay costs.
//
//   hipcc --offload-arch=gfx950 -O2 scripts/issue_probe.hip -o build/issue_probe && build/issue_probe > profiles/issue_probe_r05.jsonl
//
// Every launch: 256 x WG_PER_CU workgroups of W waves (one round: every CU holds WG_PER_CU workgroups), every wave
// ITER x 256 instructions of its role.  Reported: wall time (hipEvents), shader cycles per wave (s_memtime), the
// rate in instructions per SIMD per cycle, the SIMD ids the waves of workgroup 0 reported (HW_ID).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef uint32_t u32;
typedef uint64_t u64;

enum Mode {
    M_VALU = 0,        // 4 independent v_add chains
    M_SALU = 1,        // 4 independent s_add chains
    M_MIX = 2,         // one wave: v_add, s_add alternating (independent of each other)
    M_SPLIT = 3,       // waves in even slots of a SIMD: VALU loop, odd slots: SALU loop
    M_VDEP = 4,        // one dependent v_add chain
    M_SDEP = 5,        // one dependent s_add chain
    M_LDS_RMW = 6,     // ds_read -> wait -> v_add -> ds_write, own address (no conflicts), dependent through memory
    M_LDS_RMW_RND = 7, // the same on pseudo-random addresses of an 8 KiB window (the overlay's conflicts)
    M_RL_CHAIN = 8,    // v_readlane -> s_add -> v_add (uses the sgpr) chain: VALU->SALU->VALU forwarding
    M_LDS_RD2 = 9,     // two independent ds_reads in flight per step, then the adds and the writes
    M_VALU_LDS = 10,   // even slots VALU loop, odd slots LDS RMW: does the overlay's LDS chain hide behind VALU?
    M_MIX_DEP = 11,    // v_readlane s,v ; s_and s,s ; v_add v,v,s : the kernel's 1:1 pattern with real dependences
    M_LDS_RWW4_RND = 12,   // 4 random reads in flight, wait, 4 adds, 4 writes, 4 more writes (read / write / restore of 4 overlay halves, unchained)
    M_LDS_ATOM4_RND = 13,  // 4 random ds_add_rtn_u32 in flight, wait, 4 adds, 4 ds_sub_u32 (the fixed-point overlay: 2 LDS ops per half)
    M_LDS_ATOM4_NORTN = 14,// 4 random ds_add_u32 (no return), 4 ds_sub_u32
    M_LDS_FATOM4_RND = 15, // 4 random ds_add_rtn_f32 in flight, wait, 4 adds, 4 ds_write_b32 (the float overlay with LDS atomics)
    M_COUNT
};
static const char* mode_name[M_COUNT] = {"valu", "salu", "mix_1wave", "split_valu_salu", "valu_dep", "salu_dep", "lds_rmw",
                                          "lds_rmw_random", "readlane_salu_valu_chain", "lds_rmw_2_in_flight",
                                          "split_valu_ldsrmw", "mix_dependent", "lds_read4_write4_write4_random",
                                          "lds_addrtn4_sub4_random", "lds_add4_sub4_noreturn_random", "lds_faddrtn4_write4_random"};

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

template <int MODE>
__global__ void __launch_bounds__(1024) probe(u64* __restrict__ out, u32* __restrict__ hwid, int iters, u32 seed) {
    constexpr bool USES_LDS = MODE >= M_LDS_RWW4_RND || MODE == M_LDS_RMW || MODE == M_LDS_RMW_RND || MODE == M_LDS_RD2 || MODE == M_VALU_LDS;
    constexpr u32 NWIN = MODE >= M_LDS_RWW4_RND ? 8u : 16u;    // (the 4-half patterns: two waves share an 8 KiB window, so that 2 workgroups = 8 waves per SIMD fit a CU)
    __shared__ u32 lds[USES_LDS ? NWIN * 2048 : 64];
    const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    u32 hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    const u32 slot = hw & 0xFu;
    u32 a = lane, b = lane * 3u, c = lane + 7u, d = lane ^ 5u, one = seed | 1u;
    u32 sa = seed, sb = seed + 1u, sc = seed + 2u, sd = seed + 3u, sone = seed | 1u;
    // LDS window of this wave: 2048 words
    u32* win = lds + (USES_LDS ? (wave & (NWIN - 1u)) * 2048u : 0u);
    for (u32 i = lane; i < (USES_LDS ? 2048u : 64u); i += 64u) win[i] = i;
    typedef __attribute__((address_space(3))) u32* lds_ptr;
    u32 addr0 = (u32)(uintptr_t)(lds_ptr)(win + lane);         // byte address inside LDS
    // pseudo-random word of the window per lane and step
    u32 rnd = lane * 2654435761u + seed;
    __syncthreads();
    const u64 t0 = __builtin_readcyclecounter();
    int role = 0;
    if (MODE == M_SPLIT || MODE == M_VALU_LDS) role = (int)(slot & 1u);
    for (int it = 0; it < iters; it++) {
        if (MODE == M_VALU || ((MODE == M_SPLIT || MODE == M_VALU_LDS) && role == 0)) {
            asm volatile(REP64("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(one));
        } else if (MODE == M_SALU || (MODE == M_SPLIT && role == 1)) {
            asm volatile(REP64("s_add_u32 %0, %0, %4\n s_add_u32 %1, %1, %4\n s_add_u32 %2, %2, %4\n s_add_u32 %3, %3, %4\n")
                         : "+s"(sa), "+s"(sb), "+s"(sc), "+s"(sd) : "s"(sone) : "scc");
        } else if (MODE == M_MIX) {
            asm volatile(REP64("v_add_u32 %0, %0, %4\n s_add_u32 %2, %2, %5\n v_add_u32 %1, %1, %4\n s_add_u32 %3, %3, %5\n")
                         : "+v"(a), "+v"(b), "+s"(sa), "+s"(sb) : "v"(one), "s"(sone) : "scc");
        } else if (MODE == M_VDEP) {
            asm volatile(REP64(REP4("v_add_u32 %0, %0, %1\n")) : "+v"(a) : "v"(one));
        } else if (MODE == M_SDEP) {
            asm volatile(REP64(REP4("s_add_u32 %0, %0, %1\n")) : "+s"(sa) : "s"(sone) : "scc");
        } else if (MODE == M_LDS_RMW || (MODE == M_VALU_LDS && role == 1)) {
            // 64 round trips x 4 instructions (read, wait, add, write) = 256 "instructions"
            asm volatile(REP64("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)\n v_add_u32 %0, %0, %2\n ds_write_b32 %1, %0\n")
                         : "+v"(a) : "v"(addr0), "v"(one) : "memory");
        } else if (MODE == M_LDS_RMW_RND) {
            u32 base = (u32)(uintptr_t)(lds_ptr)win;
            asm volatile(REP64("v_mad_u32_u24 %2, %2, 5, 1\n v_and_b32 %3, 0x1ffc, %2\n v_add_u32 %3, %3, %5\n"
                               "ds_read_b32 %0, %3\n s_waitcnt lgkmcnt(0)\n v_add_u32 %0, %0, %4\n ds_write_b32 %3, %0\n")
                         : "+v"(a), "+v"(b), "+v"(rnd), "+v"(c) : "v"(one), "v"(base) : "memory");
        } else if (MODE == M_RL_CHAIN) {
            asm volatile(REP64("v_readlane_b32 %1, %0, 3\n s_add_u32 %1, %1, %2\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %3\n")
                         : "+v"(a), "+s"(sa) : "s"(sone), "v"(one) : "scc");
        } else if (MODE == M_LDS_RD2) {
            u32 addr1 = addr0 + 256u;
            asm volatile(REP16(REP4("ds_read_b32 %0, %2\n ds_read_b32 %1, %3\n s_waitcnt lgkmcnt(0)\n v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n"
                                    "ds_write_b32 %2, %0\n ds_write_b32 %3, %1\n v_add_u32 %0, %0, %4\n"))
                         : "+v"(a), "+v"(b) : "v"(addr0), "v"(addr1), "v"(one) : "memory");
        } else if (MODE == M_LDS_RWW4_RND || MODE == M_LDS_ATOM4_RND || MODE == M_LDS_ATOM4_NORTN || MODE == M_LDS_FATOM4_RND) {
            // 16 rounds of 4 halves: 4 address updates (3 VALU each), then the LDS pattern
            const u32 base = (u32)(uintptr_t)(lds_ptr)win;
            u32 r0 = rnd, r1 = rnd * 3u + 1u, r2 = rnd * 5u + 2u, r3 = rnd * 7u + 3u, a0, a1, a2, a3, v0 = 0, v1 = 0, v2 = 0, v3 = 0;
#define SA_ADDR4 "v_mad_u32_u24 %4, %4, 5, 1\n v_and_b32 %8, 0x1ffc, %4\n v_add_u32 %8, %8, %13\n" \
                 "v_mad_u32_u24 %5, %5, 5, 1\n v_and_b32 %9, 0x1ffc, %5\n v_add_u32 %9, %9, %13\n" \
                 "v_mad_u32_u24 %6, %6, 5, 1\n v_and_b32 %10, 0x1ffc, %6\n v_add_u32 %10, %10, %13\n" \
                 "v_mad_u32_u24 %7, %7, 5, 1\n v_and_b32 %11, 0x1ffc, %7\n v_add_u32 %11, %11, %13\n"
            if (MODE == M_LDS_RWW4_RND) {
                asm volatile(REP16(SA_ADDR4
                                   "ds_read_b32 %0, %8\n ds_read_b32 %1, %9\n ds_read_b32 %2, %10\n ds_read_b32 %3, %11\n s_waitcnt lgkmcnt(0)\n"
                                   "v_add_u32 %0, %0, %12\n v_add_u32 %1, %1, %12\n v_add_u32 %2, %2, %12\n v_add_u32 %3, %3, %12\n"
                                   "ds_write_b32 %8, %0\n ds_write_b32 %9, %1\n ds_write_b32 %10, %2\n ds_write_b32 %11, %3\n"
                                   "ds_write_b32 %11, %3\n ds_write_b32 %10, %2\n ds_write_b32 %9, %1\n ds_write_b32 %8, %0\n")
                             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3)
                             : "v"(one), "v"(base) : "memory");
            } else if (MODE == M_LDS_ATOM4_RND) {
                asm volatile(REP16(SA_ADDR4
                                   "ds_add_rtn_u32 %0, %8, %12\n ds_add_rtn_u32 %1, %9, %12\n ds_add_rtn_u32 %2, %10, %12\n ds_add_rtn_u32 %3, %11, %12\n s_waitcnt lgkmcnt(0)\n"
                                   "v_add_u32 %0, %0, %12\n v_add_u32 %1, %1, %12\n v_add_u32 %2, %2, %12\n v_add_u32 %3, %3, %12\n"
                                   "ds_sub_u32 %8, %12\n ds_sub_u32 %9, %12\n ds_sub_u32 %10, %12\n ds_sub_u32 %11, %12\n")
                             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3)
                             : "v"(one), "v"(base) : "memory");
            } else if (MODE == M_LDS_FATOM4_RND) {
                asm volatile(REP16(SA_ADDR4
                                   "ds_add_rtn_f32 %0, %8, %12\n ds_add_rtn_f32 %1, %9, %12\n ds_add_rtn_f32 %2, %10, %12\n ds_add_rtn_f32 %3, %11, %12\n s_waitcnt lgkmcnt(0)\n"
                                   "v_add_f32 %0, %0, %12\n v_add_f32 %1, %1, %12\n v_add_f32 %2, %2, %12\n v_add_f32 %3, %3, %12\n"
                                   "ds_write_b32 %11, %3\n ds_write_b32 %10, %2\n ds_write_b32 %9, %1\n ds_write_b32 %8, %0\n")
                             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3)
                             : "v"(one), "v"(base) : "memory");
            } else {
                asm volatile(REP16(SA_ADDR4
                                   "ds_add_u32 %8, %12\n ds_add_u32 %9, %12\n ds_add_u32 %10, %12\n ds_add_u32 %11, %12\n"
                                   "v_add_u32 %0, %0, %12\n v_add_u32 %1, %1, %12\n v_add_u32 %2, %2, %12\n v_add_u32 %3, %3, %12\n"
                                   "ds_sub_u32 %8, %12\n ds_sub_u32 %9, %12\n ds_sub_u32 %10, %12\n ds_sub_u32 %11, %12\n")
                             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3)
                             : "v"(one), "v"(base) : "memory");
            }
            a += v0 + v1 + v2 + v3;
        } else if (MODE == M_MIX_DEP) {
            asm volatile(REP64("v_readlane_b32 %1, %0, 3\n s_and_b32 %1, %1, 0xffff\n v_add_u32 %0, %0, %1\n s_add_u32 %2, %2, %3\n")
                         : "+v"(a), "+s"(sa), "+s"(sb) : "s"(sone) : "scc");
        }
    }
    const u64 t1 = __builtin_readcyclecounter();
    a += b + c + d + sa + sb + sc + sd;
    if (a == 0x12345u) win[lane] = a;                           // keep everything alive
    if (lane == 0) {
        out[(u64)blockIdx.x * (blockDim.x >> 6) + wave] = t1 - t0;
        if (blockIdx.x == 0) hwid[wave] = hw;
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE>
static void run(int waves_per_wg, int wg_per_cu, int iters, int n_cu, u64* d_out, u32* d_hw) {
    const int grid = n_cu * wg_per_cu, block = waves_per_wg * 64;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(block), 0, 0, d_out, d_hw, 2, 1u);            // warm
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; r++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(block), 0, 0, d_out, d_hw, iters, 1u);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    std::vector<u64> cyc((size_t)grid * waves_per_wg);
    std::vector<u32> hw(16);
    CK(hipMemcpy(cyc.data(), d_out, cyc.size() * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hw.data(), d_hw, 16 * 4, hipMemcpyDeviceToHost));
    double sum = 0; u64 mx = 0;
    for (u64 c : cyc) { sum += (double)c; if (c > mx) mx = c; }
    const double mean = sum / (double)cyc.size();
    const double insts = (double)iters * 256.0;
    const int waves_per_simd = waves_per_wg * wg_per_cu / 4;
    char simds[128]; int n = 0;
    for (int w = 0; w < waves_per_wg && w < 16; w++) n += snprintf(simds + n, sizeof(simds) - n, "%s%u", w ? "," : "", (hw[w] >> 4) & 3u);
    // s_memtime counts at a constant 100 MHz on this part: shader-clock cycles come from the wall time at 2.4 GHz
    printf("{\"mode\": \"%s\", \"waves_per_wg\": %d, \"wg_per_cu\": %d, \"waves_per_simd\": %d, \"inst_per_wave\": %.0f, \"ms\": %.4f, "
           "\"counter_ticks_per_wave_mean\": %.0f, \"counter_ticks_per_wave_max\": %llu, \"ns_per_inst_per_wave\": %.3f, "
           "\"cycles_at_2p4GHz_per_inst_per_wave\": %.2f, \"inst_per_simd_per_4cycles\": %.3f, \"simd_of_wave\": \"%s\"}\n",
           mode_name[MODE], waves_per_wg, wg_per_cu, waves_per_simd, insts, best, mean, (unsigned long long)mx,
           best * 1e6 / insts, best * 1e6 / insts * 2.4, waves_per_simd * insts / (best * 1e6 * 2.4) * 4.0, simds);
    fflush(stdout);
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

template <int MODE>
static void sweep(int iters, int n_cu, u64* d_out, u32* d_hw, bool small_only = false) {
    run<MODE>(4, 1, iters, n_cu, d_out, d_hw);      // 1 wave per SIMD
    run<MODE>(8, 1, iters, n_cu, d_out, d_hw);      // 2
    run<MODE>(16, 1, iters, n_cu, d_out, d_hw);     // 4
    if (!small_only) run<MODE>(16, 2, iters, n_cu, d_out, d_hw);     // 8 (LDS-window kernels: 2 x 128 KiB do not fit -> skipped via small_only)
}

int main(int argc, char** argv) {
    int iters = argc > 1 ? atoi(argv[1]) : 400;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    printf("{\"device\": \"%s\", \"arch\": \"%s\", \"cus\": %d, \"clock_khz\": %d}\n", prop.name, prop.gcnArchName, n_cu, prop.clockRate);
    u64* d_out; u32* d_hw;
    CK(hipMalloc(&d_out, (size_t)n_cu * 2 * 16 * 8));
    CK(hipMalloc(&d_hw, 16 * 4));
    sweep<M_VALU>(iters, n_cu, d_out, d_hw);
    sweep<M_SALU>(iters, n_cu, d_out, d_hw);
    sweep<M_MIX>(iters, n_cu, d_out, d_hw);
    sweep<M_SPLIT>(iters, n_cu, d_out, d_hw);
    sweep<M_VDEP>(iters, n_cu, d_out, d_hw);
    sweep<M_SDEP>(iters, n_cu, d_out, d_hw);
    sweep<M_RL_CHAIN>(iters, n_cu, d_out, d_hw);
    sweep<M_MIX_DEP>(iters, n_cu, d_out, d_hw);
    sweep<M_LDS_RMW>(iters / 4 + 1, n_cu, d_out, d_hw, true);
    sweep<M_LDS_RMW_RND>(iters / 4 + 1, n_cu, d_out, d_hw, true);
    sweep<M_LDS_RD2>(iters / 4 + 1, n_cu, d_out, d_hw, true);
    sweep<M_VALU_LDS>(iters / 4 + 1, n_cu, d_out, d_hw, true);
    // (these three: one "instruction" of the report = 1/16 of a round of 4 halves, i.e. cycles per round = 16 x the figure)
    sweep<M_LDS_RWW4_RND>(iters / 4 + 1, n_cu, d_out, d_hw, false);
    sweep<M_LDS_ATOM4_RND>(iters / 4 + 1, n_cu, d_out, d_hw, false);
    sweep<M_LDS_ATOM4_NORTN>(iters / 4 + 1, n_cu, d_out, d_hw, false);
    sweep<M_LDS_FATOM4_RND>(iters / 4 + 1, n_cu, d_out, d_hw, false);
    return 0;
}
