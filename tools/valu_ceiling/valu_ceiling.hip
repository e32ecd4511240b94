// gfx950 issue-rate calibration (verdict r05, item 2): instructions per SIMD and shader cycle for the instruction classes of the traversal loop, at 1 / 2 / 4 / 7 / 8 waves per
// SIMD, as 8 independent chains per wave (throughput) and as 1 dependent chain (latency). Build: hipcc --offload-arch=gfx950 -O2 -o valu_ceiling valu_ceiling.hip ; run on an MI355X.
// Method: every wave runs ITER x 128 instructions of one class (inline asm, registers chosen so that chain i only depends on chain i) between two s_memtime reads (tick = shader cycle,
// MI355X_MICROARCH.md) and records HW_ID; per SIMD the rate is (instructions of all its waves) / (last end - first start). Blocks of 256 threads put one wave on each SIMD of a CU; the
// dynamic LDS request (160 KB / W) makes exactly W blocks fit a CU, and the grid is 256 x W x 2 blocks so that every CU holds its W blocks for most of the run (waves that ran on an
// under-filled SIMD are filtered by the census: only SIMDs whose wave count is a multiple of W x 2 with full overlap are used; the median over SIMDs is reported).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <map>
#include <string>

#define ITER 1500
#define REP 16

struct Rec { unsigned long long t0, t1; unsigned hwid, xcc; };

__device__ __forceinline__ unsigned long long memtime() { unsigned long long t; asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }
__device__ __forceinline__ unsigned hwid() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v)); return v; }
__device__ __forceinline__ unsigned xccid() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v; }

// one kernel per class. I(r) is the instruction text acting on chain register r ("%0" .. "%7"; %8 and %9 are two loop-invariant operands, %10 an LDS address); the eight
// 128 instructions of a loop iteration sit in ONE asm statement (.rept 16), so the compiler adds nothing between them (it pads every asm statement with an s_nop)
#define ASM8(I, R0, R1, R2, R3, R4, R5, R6, R7) asm volatile(".rept 16\n" I(R0) I(R1) I(R2) I(R3) I(R4) I(R5) I(R6) I(R7) ".endr\n" \
    : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b), "v"(c), "v"(laddr) : "vcc", "scc", "s20", "s21", "s22", "s23", "memory");
#define KERNEL2(NAME, TYPE, I) \
template <int CH> __global__ void __launch_bounds__(256) NAME(Rec* rec, TYPE seed, TYPE* sink) { \
    extern __shared__ unsigned lds[]; \
    TYPE a[8]; TYPE b = seed, c = seed + (TYPE)1; \
    _Pragma("unroll") for (int k = 0; k < 8; k++) a[k] = seed + (TYPE)(threadIdx.x + k); \
    lds[threadIdx.x] = threadIdx.x; __syncthreads(); \
    unsigned laddr = (threadIdx.x & 63u) * 8u; \
    const unsigned long long t0 = memtime(); \
    _Pragma("nounroll") for (int it = 0; it < ITER; it++) { \
        { \
            if (CH == 8) { ASM8(I, "%0", "%1", "%2", "%3", "%4", "%5", "%6", "%7") } else { ASM8(I, "%0", "%0", "%0", "%0", "%0", "%0", "%0", "%0") } \
        } \
    } \
    const unsigned long long t1 = memtime(); \
    TYPE s = 0; _Pragma("unroll") for (int k = 0; k < 8; k++) s += a[k]; \
    if (s == (TYPE)123456789) sink[0] = s + b + c; \
    if ((threadIdx.x & 63u) == 0u) { Rec r; r.t0 = t0; r.t1 = t1; r.hwid = hwid(); r.xcc = xccid(); rec[blockIdx.x * 4u + (threadIdx.x >> 6)] = r; } \
}
typedef float f32; typedef unsigned u32;
#define I_FMA(r)     "v_fma_f32 " r ", " r ", %8, %9\n"
#define I_MUL(r)     "v_mul_f32 " r ", " r ", %8\n"
#define I_MAX(r)     "v_max_f32 " r ", " r ", %8\n"
#define I_MIN3(r)    "v_min3_f32 " r ", " r ", %8, %9\n"
#define I_ADDU(r)    "v_add_u32 " r ", " r ", %8\n"
#define I_AND(r)     "v_and_b32 " r ", " r ", %8\n"
#define I_LSHL(r)    "v_lshlrev_b32 " r ", 1, " r "\n"
#define I_BFE(r)     "v_bfe_u32 " r ", " r ", 8, 8\n"
#define I_SDWA(r)    "v_add_u32_sdwa " r ", " r ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
#define I_CNDMASK(r) "v_cndmask_b32 " r ", " r ", %8, vcc\n"
#define I_PERM(r)    "v_perm_b32 " r ", " r ", %8, %9\n"
#define I_DPP(r)     "v_mov_b32_dpp " r ", " r " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define I_CMP(r)     "v_cmp_lt_f32 vcc, " r ", %8\n"
#define I_CMPS(r)    "v_cmp_lt_f32_e64 s[20:21], " r ", %8\n s_and_b64 s[22:23], s[20:21], exec\n"
#define I_CVTUB(r)   "v_cvt_f32_ubyte0 " r ", " r "\n"
#define I_CVTH(r)    "v_cvt_f16_f32 " r ", " r "\n"
#define I_RCP(r)     "v_rcp_f32 " r ", " r "\n"
#define I_MULLO(r)   "v_mul_lo_u32 " r ", " r ", %8\n"
#define I_MAD24(r)   "v_mad_u32_u24 " r ", " r ", %8, %9\n"
#define I_NOP(r)     "v_nop\n"
#define I_SALU(r)    "s_add_u32 s20, s20, 1\n"
#define I_FMA_SALU(r) "v_fma_f32 " r ", " r ", %8, %9\n s_add_u32 s20, s20, 1\n"
#define I_ADDF(r)    "v_add_f32 " r ", " r ", %8\n"
#define I_MOV(r)     "v_mov_b32 " r ", %8\n"
#define I_MINF(r)    "v_min_f32 " r ", " r ", %8\n"
#define I_OR(r)      "v_or_b32 " r ", " r ", %8\n"
#define I_XOR(r)     "v_xor_b32 " r ", " r ", %8\n"
#define I_LSHLADD(r) "v_lshl_add_u32 " r ", " r ", 2, %8\n"
#define I_ADD3(r)    "v_add3_u32 " r ", " r ", %8, %9\n"
#define I_FMAC(r)    "v_fmac_f32 " r ", %8, %9\n"
#define I_SUBU(r)    "v_sub_u32 " r ", " r ", %8\n"
#define I_CVTFU(r)   "v_cvt_f32_u32 " r ", " r "\n"
#define I_CNDS(r)    "v_cndmask_b32_e64 " r ", " r ", %8, s[20:21]\n"
#define I_MUL_CND(r) "v_mul_f32 " r ", " r ", %8\n v_cndmask_b32 " r ", " r ", %9, vcc\n"
#define I_MUL_NOP(r) "v_mul_f32 " r ", " r ", %8\n v_nop\n"
#define I_CMP_CND(r) "v_cmp_lt_f32 vcc, " r ", %8\n v_cndmask_b32 " r ", " r ", %9, vcc\n"
#define I_MUL_MAX(r) "v_mul_f32 " r ", " r ", %8\n v_max_f32 " r ", " r ", %9\n"
#define I_MUL_PERM(r) "v_mul_f32 " r ", " r ", %8\n v_perm_b32 " r ", " r ", %8, %9\n"
#define I_SAND(r)    "s_and_b64 s[22:23], s[20:21], exec\n"
#define I_BALLOT(r)  "v_cmp_ne_u32_e64 s[20:21], " r ", %8\n"
#define I_RDFL(r)    "v_readfirstlane_b32 s20, " r "\n"
#define I_DSREAD(r)  "ds_read_b32 " r ", %10\n s_waitcnt lgkmcnt(0)\n"
#define I_DSREADQ(r) "ds_read_b32 " r ", %10\n"
#define I_DSREAD64(r) "ds_read_b32 " r ", %10 offset:64\n"
KERNEL2(k_fma, f32, I_FMA) KERNEL2(k_mul, f32, I_MUL) KERNEL2(k_max, f32, I_MAX) KERNEL2(k_min3, f32, I_MIN3) KERNEL2(k_addu, u32, I_ADDU) KERNEL2(k_and, u32, I_AND) KERNEL2(k_lshl, u32, I_LSHL)
KERNEL2(k_bfe, u32, I_BFE) KERNEL2(k_sdwa, u32, I_SDWA) KERNEL2(k_cndmask, u32, I_CNDMASK) KERNEL2(k_perm, u32, I_PERM) KERNEL2(k_dpp, u32, I_DPP) KERNEL2(k_cmp, f32, I_CMP) KERNEL2(k_cmps, f32, I_CMPS)
KERNEL2(k_cvtub, u32, I_CVTUB) KERNEL2(k_cvth, f32, I_CVTH) KERNEL2(k_rcp, f32, I_RCP) KERNEL2(k_mullo, u32, I_MULLO) KERNEL2(k_mad24, u32, I_MAD24) KERNEL2(k_nop, u32, I_NOP) KERNEL2(k_salu, u32, I_SALU)
KERNEL2(k_fma_salu, f32, I_FMA_SALU) KERNEL2(k_dsread, u32, I_DSREAD)
KERNEL2(k_addf, f32, I_ADDF) KERNEL2(k_mov, u32, I_MOV) KERNEL2(k_minf, f32, I_MINF) KERNEL2(k_or, u32, I_OR) KERNEL2(k_xor, u32, I_XOR) KERNEL2(k_lshladd, u32, I_LSHLADD) KERNEL2(k_add3, u32, I_ADD3)
KERNEL2(k_fmac, f32, I_FMAC) KERNEL2(k_subu, u32, I_SUBU) KERNEL2(k_cvtfu, u32, I_CVTFU) KERNEL2(k_cnds, u32, I_CNDS) KERNEL2(k_mul_cnd, f32, I_MUL_CND) KERNEL2(k_mul_nop, f32, I_MUL_NOP) KERNEL2(k_cmp_cnd, f32, I_CMP_CND)
KERNEL2(k_mul_max, f32, I_MUL_MAX) KERNEL2(k_mul_perm, u32, I_MUL_PERM) KERNEL2(k_sand, u32, I_SAND) KERNEL2(k_ballot, u32, I_BALLOT) KERNEL2(k_rdfl, u32, I_RDFL)
// eight LDS reads in flight, one wait
template <int CH> __global__ void __launch_bounds__(256) k_dsreadq(Rec* rec, u32 seed, u32* sink) {
    extern __shared__ unsigned lds[];
    u32 a[8]; u32 b = seed, c = seed + 1u;
    for (int k = 0; k < 8; k++) a[k] = seed + threadIdx.x + k;
    lds[threadIdx.x] = threadIdx.x; __syncthreads();
    unsigned laddr = (threadIdx.x & 63u) * 8u;
    const unsigned long long t0 = memtime();
#pragma nounroll
    for (int it = 0; it < ITER; it++) {
        {
            asm volatile(".rept 16\n" I_DSREADQ("%0") I_DSREADQ("%1") I_DSREADQ("%2") I_DSREADQ("%3") I_DSREADQ("%4") I_DSREADQ("%5") I_DSREADQ("%6") I_DSREADQ("%7") "s_waitcnt lgkmcnt(0)\n.endr\n"
                : "=v"(a[0]), "=v"(a[1]), "=v"(a[2]), "=v"(a[3]), "=v"(a[4]), "=v"(a[5]), "=v"(a[6]), "=v"(a[7]) : "v"(b), "v"(c), "v"(laddr) : "memory");
        }
    }
    const unsigned long long t1 = memtime();
    u32 s = 0; for (int k = 0; k < 8; k++) s += a[k];
    if (s == 123456789u) sink[0] = s + b + c;
    if ((threadIdx.x & 63u) == 0u) { Rec r; r.t0 = t0; r.t1 = t1; r.hwid = hwid(); r.xcc = xccid(); rec[blockIdx.x * 4u + (threadIdx.x >> 6)] = r; }
}

// packed fp32 needs register pairs
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define PK8(I, R0, R1, R2, R3, R4, R5, R6, R7) asm volatile(".rept 16\n" I(R0) I(R1) I(R2) I(R3) I(R4) I(R5) I(R6) I(R7) ".endr\n" \
    : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b), "v"(c));
#define PK_KERNEL(NAME, I) \
template <int CH> __global__ void __launch_bounds__(256) NAME(Rec* rec, float seed, float* sink) { \
    extern __shared__ unsigned lds[]; \
    f32x2 a[8]; f32x2 b = {seed, seed + 1.f}, c = {seed + 2.f, seed}; \
    _Pragma("unroll") for (int k = 0; k < 8; k++) a[k] = (f32x2){seed + threadIdx.x, seed + k}; \
    lds[threadIdx.x] = threadIdx.x; __syncthreads(); \
    const unsigned long long t0 = memtime(); \
    _Pragma("nounroll") for (int it = 0; it < ITER; it++) { \
        { \
            if (CH == 8) { PK8(I, "%0", "%1", "%2", "%3", "%4", "%5", "%6", "%7") } else { PK8(I, "%0", "%0", "%0", "%0", "%0", "%0", "%0", "%0") } \
        } \
    } \
    const unsigned long long t1 = memtime(); \
    float s = 0; _Pragma("unroll") for (int k = 0; k < 8; k++) s += a[k].x + a[k].y; \
    if (s == 123456789.f) sink[0] = s; \
    if ((threadIdx.x & 63u) == 0u) { Rec r; r.t0 = t0; r.t1 = t1; r.hwid = hwid(); r.xcc = xccid(); rec[blockIdx.x * 4u + (threadIdx.x >> 6)] = r; } \
}
#define I_PKFMA(r) "v_pk_fma_f32 " r ", " r ", %8, %9\n"
#define I_PKMUL(r) "v_pk_mul_f32 " r ", " r ", %8\n"
#define I_PKADD(r) "v_pk_add_f32 " r ", " r ", %8\n"
PK_KERNEL(k_pkfma, I_PKFMA) PK_KERNEL(k_pkmul, I_PKMUL) PK_KERNEL(k_pkadd, I_PKADD)

struct Result { double rate, cyclesPerInstrOneWave, ghz; unsigned simds, simdsUsed; };

// One round of blocks: CUs x W blocks of 256 threads, W blocks per CU by LDS size. The census (XCC_ID, HW_ID: SE, SH, CU, SIMD) says how many waves every SIMD really held; a SIMD's rate
// is (instructions of its waves) / (its last end - its first start), timestamps of one XCD being one counter. Reported: the median over the SIMDs that held exactly W waves.
template <class K, class T> static Result run(K kern, T seed, int W, unsigned instrPerBody) {
    int cus = 0; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const unsigned blocks = (unsigned)cus * (unsigned)W;
    size_t ldsBytes = (size_t)(160 * 1024 / W) - 1024; if (ldsBytes > 159 * 1024) ldsBytes = 159 * 1024;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes);
    Rec* d; hipMalloc(&d, sizeof(Rec) * blocks * 4); T* sink; hipMalloc(&sink, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), ldsBytes, 0, d, seed, sink);      // warm-up (clocks, code)
    hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), ldsBytes, 0, d, seed, sink); hipEventRecord(e1);
    hipDeviceSynchronize();
    if (hipGetLastError() != hipSuccess) { fprintf(stderr, "launch failed\n"); exit(1); }
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<Rec> h(blocks * 4); hipMemcpy(h.data(), d, sizeof(Rec) * h.size(), hipMemcpyDeviceToHost);
    const double instrPerWave = (double)ITER * 128 * instrPerBody;
    struct Simd { unsigned n = 0; unsigned long long t0 = ~0ull, t1 = 0; };
    std::map<unsigned long long, Simd> simds;
    double sumSpan = 0;
    for (auto& r : h) {
        const unsigned long long key = ((unsigned long long)(r.xcc & 0xFu) << 32) | (r.hwid & 0xFFF0u);      // HW_ID: wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13]
        Simd& s = simds[key]; s.n++; s.t0 = std::min(s.t0, r.t0); s.t1 = std::max(s.t1, r.t1); sumSpan += (double)(r.t1 - r.t0);
    }
    std::vector<double> rates; double spanSum = 0;
    for (auto& kv : simds) if (kv.second.n == (unsigned)W) { rates.push_back(kv.second.n * instrPerWave / (double)(kv.second.t1 - kv.second.t0)); spanSum += (double)(kv.second.t1 - kv.second.t0); }
    std::sort(rates.begin(), rates.end());
    Result R; R.simds = (unsigned)simds.size(); R.simdsUsed = (unsigned)rates.size();
    R.rate = rates.empty() ? 0.0 : rates[rates.size() / 2];
    R.cyclesPerInstrOneWave = sumSpan / h.size() / instrPerWave;
    R.ghz = rates.empty() ? 0.0 : (spanSum / rates.size()) / (ms * 1e-3) / 1e9;      // a SIMD's busy span in ticks over the launch's wall time: a lower bound of the tick rate
    hipFree(d); hipFree(sink); hipEventDestroy(e0); hipEventDestroy(e1);
    return R;
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("# device %s, %d CUs, clockRate %d kHz; ITER %d x %d instructions per wave (one asm block of 128 per loop iteration); blocks of 256 threads (one wave per SIMD), W blocks per CU by LDS size\n", p.gcnArchName, p.multiProcessorCount, p.clockRate, ITER, 128);
    printf("# rate = instructions per SIMD and shader cycle (s_memtime ticks); 8 chains = independent (throughput), 1 chain = every instruction depends on the one before (latency)\n");
    printf("%-34s %6s | %-47s | %-47s | %s\n", "class", "", "8 chains: rate at W = 1 / 2 / 4 / 7 / 8 waves/SIMD", "1 chain: rate at W = 1 / 2 / 4 / 7 / 8", "cycles/instr of ONE wave alone (8 ch / 1 ch); ticks per second >= (W = 8); SIMDs with exactly 8 waves / SIMDs seen");
    const int Ws[5] = {1, 2, 4, 7, 8};
#define ROW(LABEL, K8, K1, SEED, IPB) { double r8[5], r1[5], c8 = 0, c1 = 0, ghz = 0; unsigned used = 0, tot = 0; \
        for (int i = 0; i < 5; i++) { Result a = run(K8, SEED, Ws[i], IPB); Result b = run(K1, SEED, Ws[i], IPB); r8[i] = a.rate; r1[i] = b.rate; if (i == 0) { c8 = a.cyclesPerInstrOneWave; c1 = b.cyclesPerInstrOneWave; } if (i == 4) { ghz = a.ghz; used = a.simdsUsed; tot = a.simds; } } \
        printf("%-34s %6s | %8.3f %8.3f %8.3f %8.3f %8.3f    | %8.3f %8.3f %8.3f %8.3f %8.3f    | %6.2f / %6.2f   %.2f GHz  %u/%u\n", LABEL, "", r8[0], r8[1], r8[2], r8[3], r8[4], r1[0], r1[1], r1[2], r1[3], r1[4], c8, c1, ghz, used, tot); fflush(stdout); }
    ROW("v_fma_f32", k_fma<8>, k_fma<1>, 1.0f, 1)
    ROW("v_mul_f32", k_mul<8>, k_mul<1>, 1.0f, 1)
    ROW("v_add_f32", k_addf<8>, k_addf<1>, 1.0f, 1)
    ROW("v_fmac_f32", k_fmac<8>, k_fmac<1>, 1.0f, 1)
    ROW("v_min_f32", k_minf<8>, k_minf<1>, 1.0f, 1)
    ROW("v_mov_b32", k_mov<8>, k_mov<1>, 1u, 1)
    ROW("v_max_f32", k_max<8>, k_max<1>, 1.0f, 1)
    ROW("v_min3_f32", k_min3<8>, k_min3<1>, 1.0f, 1)
    ROW("v_pk_fma_f32", k_pkfma<8>, k_pkfma<1>, 1.0f, 1)
    ROW("v_pk_mul_f32", k_pkmul<8>, k_pkmul<1>, 1.0f, 1)
    ROW("v_pk_add_f32", k_pkadd<8>, k_pkadd<1>, 1.0f, 1)
    ROW("v_add_u32", k_addu<8>, k_addu<1>, 1u, 1)
    ROW("v_and_b32", k_and<8>, k_and<1>, 0xFFFFFFFFu, 1)
    ROW("v_sub_u32", k_subu<8>, k_subu<1>, 1u, 1)
    ROW("v_or_b32", k_or<8>, k_or<1>, 1u, 1)
    ROW("v_xor_b32", k_xor<8>, k_xor<1>, 1u, 1)
    ROW("v_lshl_add_u32", k_lshladd<8>, k_lshladd<1>, 1u, 1)
    ROW("v_add3_u32", k_add3<8>, k_add3<1>, 1u, 1)
    ROW("v_cvt_f32_u32", k_cvtfu<8>, k_cvtfu<1>, 1u, 1)
    ROW("v_lshlrev_b32", k_lshl<8>, k_lshl<1>, 1u, 1)
    ROW("v_bfe_u32", k_bfe<8>, k_bfe<1>, 1u, 1)
    ROW("v_add_u32_sdwa (byte select)", k_sdwa<8>, k_sdwa<1>, 1u, 1)
    ROW("v_cndmask_b32 (vcc)", k_cndmask<8>, k_cndmask<1>, 1u, 1)
    ROW("v_cndmask_b32_e64 (sgpr pair)", k_cnds<8>, k_cnds<1>, 1u, 1)
    ROW("v_mul_f32 ; v_cndmask_b32 vcc", k_mul_cnd<8>, k_mul_cnd<1>, 1.0f, 2)
    ROW("v_cmp_lt_f32 vcc ; v_cndmask vcc", k_cmp_cnd<8>, k_cmp_cnd<1>, 1.0f, 2)
    ROW("v_mul_f32 ; v_nop", k_mul_nop<8>, k_mul_nop<1>, 1.0f, 2)
    ROW("v_mul_f32 ; v_max_f32", k_mul_max<8>, k_mul_max<1>, 1.0f, 2)
    ROW("v_mul_f32 ; v_perm_b32", k_mul_perm<8>, k_mul_perm<1>, 1u, 2)
    ROW("v_cmp_ne_u32_e64 sgpr (ballot)", k_ballot<8>, k_ballot<1>, 1u, 1)
    ROW("v_readfirstlane_b32", k_rdfl<8>, k_rdfl<1>, 1u, 1)
    ROW("s_and_b64 (scalar unit)", k_sand<8>, k_sand<1>, 1u, 1)
    ROW("v_perm_b32", k_perm<8>, k_perm<1>, 0x03020100u, 1)
    ROW("v_mov_b32 dpp quad_perm", k_dpp<8>, k_dpp<1>, 1u, 1)
    ROW("v_cmp_lt_f32 vcc", k_cmp<8>, k_cmp<1>, 1.0f, 1)
    ROW("v_cmp_e64 sgpr + s_and_b64 exec", k_cmps<8>, k_cmps<1>, 1.0f, 2)
    ROW("v_cvt_f32_ubyte0", k_cvtub<8>, k_cvtub<1>, 1u, 1)
    ROW("v_cvt_f16_f32", k_cvth<8>, k_cvth<1>, 1.0f, 1)
    ROW("v_rcp_f32", k_rcp<8>, k_rcp<1>, 1.5f, 1)
    ROW("v_mul_lo_u32", k_mullo<8>, k_mullo<1>, 3u, 1)
    ROW("v_mad_u32_u24", k_mad24<8>, k_mad24<1>, 3u, 1)
    ROW("v_nop", k_nop<8>, k_nop<1>, 1u, 1)
    ROW("s_add_u32 (scalar unit)", k_salu<8>, k_salu<1>, 1u, 1)
    ROW("v_fma_f32 + s_add_u32 pairs", k_fma_salu<8>, k_fma_salu<1>, 1.0f, 2)
    ROW("ds_read_b32 + wait each", k_dsread<8>, k_dsread<1>, 1u, 1)
    ROW("ds_read_b32 x8 then wait", k_dsreadq<8>, k_dsreadq<1>, 1u, 1)
    return 0;
}
