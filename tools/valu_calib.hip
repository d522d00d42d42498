// valu_calib.hip -- gfx950 microbenchmark: SIMD issue cycles per wave64 instruction, per instruction class.
//
// Why: the "VALU issue fraction" of k_distance / k_ec_fast was priced at 4 cycles per SQ_INSTS_VALU unit on the assumption of a
// 16-lane SIMD. MI355X_MICROARCH.md says SIMD-32: 2 cycles for fp32 / int, 4 for fp64 FMA, more for the
// rcp / sqrt / div helpers. This program MEASURES the figure per class on dependency-free instruction streams, so that the busy
// fraction can be computed as   sum_class(count_class x cycles_class) / (SIMDs x duration x clock)   with the class counts taken from
// the per-class PMC counters (SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64, _INT32, _INT64, _CVT, ...; tools/pmc_traffic.py).
//
//   hipcc --offload-arch=gfx950 -O2 tools/valu_calib.hip -o tools/valu_calib && tools/valu_calib [waves_per_simd ...]
//
// Every kernel runs ITERS iterations of a loop whose body is 32 instructions of ONE opcode on 8 independent register sets (no
// instruction depends on the previous 7). A wave times its own loop with s_memtime (shader clock) and s_memrealtime (100 MHz), which
// also yields the shader clock during the run. With W waves per SIMD, cycles per instruction = W-th of a wave's loop time / 32 / ITERS.
// Output: one JSON line per (class, W): median / min over waves of the per-wave figure and the whole-launch figure from HIP events.
// Run it under `rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES ...` to get counter units per instruction as well
// (kernel names carry the class).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int ITERS = 2048, BODY = 32;

struct Stamp { unsigned long long cycles, ticks100MHz; };

// R(i) = the i-th independent register set. Each body line is one instruction; 8 sets x 4 repetitions = 32 instructions.
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define BODY32(X) REP8(X) REP8(X) REP8(X) REP8(X)

#define KERNEL_BEGIN(name) \
    __global__ void __launch_bounds__(256) name(Stamp *stamps, double *sink, int iters) { \
        double d[8], e[8]; float f[8], g[8]; int n[8], m[8]; \
        for (int i = 0; i < 8; ++i) { d[i] = 1.0+1e-9*(threadIdx.x+i); e[i] = 1.0-1e-9*(threadIdx.x+2*i); f[i] = 1.f+1e-6f*(threadIdx.x+i); g[i] = .999f; n[i] = threadIdx.x*7+i; m[i] = i+3; } \
        const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64(); \
        for (int it = 0; it < iters; ++it) {
#define KERNEL_END \
        } \
        const unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64(); \
        double s = 0; for (int i = 0; i < 8; ++i) s += d[i]+e[i]+f[i]+g[i]+n[i]+m[i]; \
        if (s == 12345.678) sink[0] = s; \
        if ((threadIdx.x&63) == 0) { Stamp st; st.cycles = t1-t0; st.ticks100MHz = w1-w0; stamps[(blockIdx.x*blockDim.x+threadIdx.x)>>6] = st; } \
    }

#define ASM1(ins, C0, V0) asm volatile(ins : C0(V0));
// ---- fp64
#define X_FMA64(i) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[i]) : "v"(e[i]));
#define X_ADD64(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(e[i]));
#define X_MUL64(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(e[i]));
#define X_MAX64(i) asm volatile("v_max_f64 %0, %0, %1" : "+v"(d[i]) : "v"(e[i]));
#define X_RCP64(i) asm volatile("v_rcp_f64 %0, %1" : "=v"(d[i]) : "v"(e[i]));
#define X_RSQ64(i) asm volatile("v_rsq_f64 %0, %1" : "=v"(d[i]) : "v"(e[i]));
#define X_SQRT64(i) asm volatile("v_sqrt_f64 %0, %1" : "=v"(d[i]) : "v"(e[i]));
#define X_DIVSCALE64(i) asm volatile("v_div_scale_f64 %0, vcc, %1, %1, %0" : "+v"(d[i]) : "v"(e[i]) : "vcc");
#define X_DIVFMAS64(i) asm volatile("v_div_fmas_f64 %0, %0, %1, %1" : "+v"(d[i]) : "v"(e[i]) : "vcc");
#define X_DIVFIXUP64(i) asm volatile("v_div_fixup_f64 %0, %0, %1, %1" : "+v"(d[i]) : "v"(e[i]));
#define X_CMP64(i) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(d[i]), "v"(e[i]) : "vcc");
#define X_CMPCLASS64(i) asm volatile("v_cmp_class_f64 vcc, %0, %1" : : "v"(d[i]), "v"(m[i]) : "vcc");
#define X_LDEXP64(i) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(d[i]) : "v"(m[i]));
#define X_FREXPM64(i) asm volatile("v_frexp_mant_f64 %0, %1" : "=v"(d[i]) : "v"(e[i]));
#define X_CVT_F32_F64(i) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(e[i]));
#define X_CVT_F64_F32(i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(g[i]));
#define X_CVT_F64_I32(i) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(d[i]) : "v"(m[i]));
// ---- fp32
#define X_FMA32(i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[i]) : "v"(g[i]));
#define X_ADD32(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(g[i]));
#define X_MUL32(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[i]) : "v"(g[i]));
#define X_RCP32(i) asm volatile("v_rcp_f32 %0, %1" : "=v"(f[i]) : "v"(g[i]));
#define X_SQRT32(i) asm volatile("v_sqrt_f32 %0, %1" : "=v"(f[i]) : "v"(g[i]));
#define X_CMP32(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(f[i]), "v"(g[i]) : "vcc");
// ---- moves / selects / integer
#define X_MOV32(i) asm volatile("v_mov_b32 %0, %1" : "=v"(n[i]) : "v"(m[i]));
#define X_CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(n[i]) : "v"(m[i]) : "vcc");
#define X_ADDU32(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(n[i]) : "v"(m[i]));
#define X_AND32(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(n[i]) : "v"(m[i]));
#define X_MULLO32(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(n[i]) : "v"(m[i]));
#define X_LSHL64(i) asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(d[i]));
#define X_MADU64(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(d[i]) : "v"(m[i]) : "vcc");
#define X_DPP(i) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(n[i]) : "v"(m[i]));
#define X_READLANE(i) asm volatile("v_readlane_b32 s20, %0, 3" : : "v"(n[i]) : "s20");
#define X_READFIRST(i) asm volatile("v_readfirstlane_b32 s20, %0" : : "v"(n[i]) : "s20");
// (c_cndmask_b32 above reads a vcc nobody wrote and measured 22 cycles -- an artefact; these are the forms the kernels execute)
#define X_CMPCND(i) asm volatile("v_cmp_lt_u32 vcc, %1, %0\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(n[i]) : "v"(m[i]) : "vcc");
#define X_CNDSGPR(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[22:23]" : "+v"(n[i]) : "v"(m[i]));
#define X_CMP64S(i) asm volatile("v_cmp_lt_f64_e64 s[24:25], %0, %1" : : "v"(d[i]), "v"(e[i]) : "s24", "s25");
#define X_MOV64(i) asm volatile("v_mov_b64 %0, %1" : "=v"(d[i]) : "v"(e[i]));
#define X_SMOV8(i) asm volatile("s_mov_b32 s2%c0, 0x3ff00000" : : "i"(i) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
#define X_SAND64(i) asm volatile("s_and_b64 s[24:25], s[26:27], exec" : : : "s24", "s25", "scc");
#define X_SNOP(i) asm volatile("s_nop 0");
// ---- scalar unit (one wave's stream; shows what SALU work costs a wave that has nothing else to issue)
#define X_SMOV(i) asm volatile("s_mov_b32 s20, 0x3ff00000" : : : "s20");
#define X_SADD(i) asm volatile("s_add_u32 s20, s20, 3" : : : "s20", "scc");
// ---- a mixed stream shaped like the distance kernel's inner loops: 2 fp64 fma, 1 fp64 mul, 1 fp64 add, 1 cmp, 1 cndmask pair, 1 s_mov
#define X_MIX(i) asm volatile("v_fma_f64 %0, %0, %2, %0\n v_mul_f64 %1, %1, %2\n s_mov_b32 s20, 0x3ff00000\n v_cmp_lt_f64 vcc, %0, %1\n v_cndmask_b32 %3, %3, %4, vcc" : "+v"(d[i]), "+v"(e[i]) : "v"(d[(i+1)&7]), "v"(n[i]), "v"(m[i]) : "vcc", "s20");

#define DEF(name, X) KERNEL_BEGIN(name) BODY32(X) KERNEL_END
DEF(c_fma_f64, X_FMA64) DEF(c_add_f64, X_ADD64) DEF(c_mul_f64, X_MUL64) DEF(c_max_f64, X_MAX64) DEF(c_rcp_f64, X_RCP64) DEF(c_rsq_f64, X_RSQ64)
DEF(c_sqrt_f64, X_SQRT64) DEF(c_div_scale_f64, X_DIVSCALE64) DEF(c_div_fmas_f64, X_DIVFMAS64) DEF(c_div_fixup_f64, X_DIVFIXUP64) DEF(c_cmp_f64, X_CMP64)
DEF(c_cmp_class_f64, X_CMPCLASS64) DEF(c_ldexp_f64, X_LDEXP64) DEF(c_frexp_mant_f64, X_FREXPM64) DEF(c_cvt_f32_f64, X_CVT_F32_F64) DEF(c_cvt_f64_f32, X_CVT_F64_F32)
DEF(c_cvt_f64_i32, X_CVT_F64_I32) DEF(c_fma_f32, X_FMA32) DEF(c_add_f32, X_ADD32) DEF(c_mul_f32, X_MUL32) DEF(c_rcp_f32, X_RCP32) DEF(c_sqrt_f32, X_SQRT32)
DEF(c_cmp_f32, X_CMP32) DEF(c_mov_b32, X_MOV32) DEF(c_cndmask_b32, X_CNDMASK) DEF(c_add_u32, X_ADDU32) DEF(c_and_b32, X_AND32) DEF(c_mul_lo_u32, X_MULLO32)
DEF(c_lshl_b64, X_LSHL64) DEF(c_mad_u64_u32, X_MADU64) DEF(c_mov_dpp, X_DPP) DEF(c_readlane, X_READLANE) DEF(c_readfirstlane, X_READFIRST)
DEF(c_s_mov, X_SMOV) DEF(c_s_add, X_SADD) DEF(c_mix5, X_MIX)
DEF(c_cmp_cndmask_pair, X_CMPCND) DEF(c_cndmask_sgprmask, X_CNDSGPR) DEF(c_cmp_f64_to_sgpr, X_CMP64S) DEF(c_mov_b64, X_MOV64) DEF(c_s_mov_8dst, X_SMOV8)
DEF(c_s_and_b64, X_SAND64) DEF(c_s_nop, X_SNOP)

// The compiler's own fp64 division and sqrt (IEEE-correct, as the product is built): 8 independent quotients per iteration.
__global__ void __launch_bounds__(256) c_ieee_div_f64(Stamp *stamps, double *sink, int iters) {
    double d[8], e[8];
    for (int i = 0; i < 8; ++i) d[i] = 1.0+1e-9*(threadIdx.x+i), e[i] = 1.0000001+1e-9*i;
    const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it)
        for (int i = 0; i < 8; ++i) d[i] = d[i]/e[i];
    const unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    double s = 0; for (int i = 0; i < 8; ++i) s += d[i];
    if (s == 12345.678) sink[0] = s;
    if ((threadIdx.x&63) == 0) { Stamp st; st.cycles = t1-t0; st.ticks100MHz = w1-w0; stamps[(blockIdx.x*blockDim.x+threadIdx.x)>>6] = st; }
}
__global__ void __launch_bounds__(256) c_ieee_sqrt_f64(Stamp *stamps, double *sink, int iters) {
    double d[8];
    for (int i = 0; i < 8; ++i) d[i] = 1e30+1e21*(threadIdx.x+i);
    const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it)
        for (int i = 0; i < 8; ++i) d[i] = sqrt(d[i])+1e30;
    const unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    double s = 0; for (int i = 0; i < 8; ++i) s += d[i];
    if (s == 12345.678) sink[0] = s;
    if ((threadIdx.x&63) == 0) { Stamp st; st.cycles = t1-t0; st.ticks100MHz = w1-w0; stamps[(blockIdx.x*blockDim.x+threadIdx.x)>>6] = st; }
}

// ---- FETCH_SIZE / WRITE_SIZE calibration (--fetch): known byte counts in the access patterns of the product's kernels, over a buffer
// larger than the 256 MiB Infinity Cache. Run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes).
__global__ void __launch_bounds__(256) f_read_16B_per_lane(const uint4 *src, size_t n16, unsigned *sink) {   // wide coalesced streaming read
    unsigned acc = 0;
    for (size_t i = (size_t) blockIdx.x*blockDim.x+threadIdx.x; i < n16; i += (size_t) gridDim.x*blockDim.x) { const uint4 v = src[i]; acc ^= v.x^v.y^v.z^v.w; }
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ void __launch_bounds__(256) f_read_4B_per_lane(const unsigned *src, size_t n4, unsigned *sink) {   // dword per lane (list / stencil reads)
    unsigned acc = 0;
    for (size_t i = (size_t) blockIdx.x*blockDim.x+threadIdx.x; i < n4; i += (size_t) gridDim.x*blockDim.x) acc ^= src[i];
    if (acc == 0x12345678u) sink[0] = acc;
}
// wave-uniform scalar loads of whole 368-byte records (k_distance's record walk): wave w reads records w, w+nWaves, ... each exactly once
__global__ void __launch_bounds__(64) f_read_sload_records(const char *src, size_t nRecords, unsigned *sink) {
    unsigned acc = 0;
    for (size_t r = blockIdx.x; r < nRecords; r += gridDim.x) {
        const char *p = src+r*368;
        unsigned a, b;
        asm volatile("s_load_dwordx16 s[36:51], %1, 0x0\n s_load_dwordx16 s[52:67], %1, 0x40\n s_load_dwordx16 s[68:83], %1, 0x80\n s_load_dwordx16 s[84:99], %1, 0xc0\n"
                     "s_waitcnt lgkmcnt(0)\n s_xor_b32 %0, s36, s99" : "=s"(a) : "s"(p)
                     : "s36","s37","s38","s39","s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51","s52","s53","s54","s55","s56","s57","s58","s59",
                       "s60","s61","s62","s63","s64","s65","s66","s67","s68","s69","s70","s71","s72","s73","s74","s75","s76","s77","s78","s79","s80","s81","s82","s83",
                       "s84","s85","s86","s87","s88","s89","s90","s91","s92","s93","s94","s95","s96","s97","s98","s99");
        asm volatile("s_load_dwordx16 s[36:51], %1, 0x100\n s_load_dwordx8 s[52:59], %1, 0x140\n s_load_dwordx4 s[60:63], %1, 0x160\n"
                     "s_waitcnt lgkmcnt(0)\n s_xor_b32 %0, s36, s63" : "=s"(b) : "s"(p)
                     : "s36","s37","s38","s39","s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51","s52","s53","s54","s55","s56","s57","s58","s59",
                       "s60","s61","s62","s63");
        acc ^= a^b;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ void __launch_bounds__(256) f_write_16B_per_lane(uint4 *dst, size_t n16) {
    for (size_t i = (size_t) blockIdx.x*blockDim.x+threadIdx.x; i < n16; i += (size_t) gridDim.x*blockDim.x) dst[i] = make_uint4((unsigned) i, 1, 2, 3);
}
// three dword stores per lane at a 12-byte stride (the msdf texel stores of k_distance): every byte of the range written once
__global__ void __launch_bounds__(256) f_write_3x4B_per_lane(float *dst, size_t nTexels) {
    for (size_t i = (size_t) blockIdx.x*blockDim.x+threadIdx.x; i < nTexels; i += (size_t) gridDim.x*blockDim.x) {
        float *px = dst+3*i;
        px[0] = (float) i; px[1] = 1.f; px[2] = 2.f;
    }
}
__global__ void __launch_bounds__(256) f_write_1B_per_lane(unsigned char *dst, size_t n) {                   // stencil bytes
    for (size_t i = (size_t) blockIdx.x*blockDim.x+threadIdx.x; i < n; i += (size_t) gridDim.x*blockDim.x) dst[i] = (unsigned char) i;
}

static int fetchCalibration() {
    const size_t bytes = (size_t) 1536<<20;                            // 1.5 GiB: six times the Infinity Cache
    char *buf; unsigned *sink;
    CHK(hipMalloc(&buf, bytes)); CHK(hipMalloc(&sink, 64));
    CHK(hipMemset(buf, 1, bytes)); CHK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    struct { const char *name; int kind; } runs[] = { { "f_read_16B_per_lane", 0 }, { "f_read_4B_per_lane", 1 }, { "f_read_sload_records", 2 },
                                                     { "f_write_16B_per_lane", 3 }, { "f_write_3x4B_per_lane", 4 }, { "f_write_1B_per_lane", 5 } };
    for (auto &r : runs) {
        size_t moved = bytes;
        CHK(hipEventRecord(e0));
        switch (r.kind) {
        case 0: f_read_16B_per_lane<<<8192, 256>>>((const uint4 *) buf, bytes/16, sink); break;
        case 1: f_read_4B_per_lane<<<8192, 256>>>((const unsigned *) buf, bytes/4, sink); break;
        case 2: moved = (bytes/368)*368; f_read_sload_records<<<65536, 64>>>(buf, bytes/368, sink); break;
        case 3: f_write_16B_per_lane<<<8192, 256>>>((uint4 *) buf, bytes/16); break;
        case 4: moved = (bytes/12)*12; f_write_3x4B_per_lane<<<8192, 256>>>((float *) buf, bytes/12); break;
        case 5: f_write_1B_per_lane<<<8192, 256>>>((unsigned char *) buf, bytes); break;
        }
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        printf("{\"case\": \"%s\", \"bytes\": %zu, \"ms\": %.3f, \"gb_per_s\": %.1f}\n", r.name, moved, ms, moved/ms*1e-6);
    }
    return 0;
}

typedef void (*Kern)(Stamp *, double *, int);
struct Case { const char *name; Kern k; int perIter; const char *pmcClass; };

int main(int argc, char **argv) {
    if (argc > 1 && !strcmp(argv[1], "--fetch"))
        return fetchCalibration();
    std::vector<int> wps;
    for (int i = 1; i < argc; ++i) wps.push_back(atoi(argv[i]));
    if (wps.empty()) wps = { 1, 2, 4, 8 };
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount, simds = cus*4;
    const Case cases[] = {
#define C(n, cls) { #n, n, BODY, cls }
        C(c_fma_f64, "FMA_F64"), C(c_add_f64, "ADD_F64"), C(c_mul_f64, "MUL_F64"), C(c_max_f64, "other"), C(c_rcp_f64, "TRANS_F64"), C(c_rsq_f64, "TRANS_F64"),
        C(c_sqrt_f64, "TRANS_F64"), C(c_div_scale_f64, "other"), C(c_div_fmas_f64, "FMA_F64?"), C(c_div_fixup_f64, "other"), C(c_cmp_f64, "other"),
        C(c_cmp_class_f64, "other"), C(c_ldexp_f64, "other"), C(c_frexp_mant_f64, "other"), C(c_cvt_f32_f64, "CVT"), C(c_cvt_f64_f32, "CVT"), C(c_cvt_f64_i32, "CVT"),
        C(c_fma_f32, "FMA_F32"), C(c_add_f32, "ADD_F32"), C(c_mul_f32, "MUL_F32"), C(c_rcp_f32, "TRANS_F32"), C(c_sqrt_f32, "TRANS_F32"), C(c_cmp_f32, "other"),
        C(c_mov_b32, "other"), C(c_cndmask_b32, "other"), C(c_add_u32, "INT32"), C(c_and_b32, "INT32"), C(c_mul_lo_u32, "INT32"), C(c_lshl_b64, "INT64"),
        C(c_mad_u64_u32, "INT64"), C(c_mov_dpp, "other"), C(c_readlane, "other"), C(c_readfirstlane, "other"), C(c_s_mov, "SALU"), C(c_s_add, "SALU"),
        { "c_cmp_cndmask_pair", c_cmp_cndmask_pair, BODY*2, "other (v_cmp_lt_u32 + v_cndmask_b32 through vcc; per instruction)" },
        C(c_cndmask_sgprmask, "other"), C(c_cmp_f64_to_sgpr, "other"), C(c_mov_b64, "other"), C(c_s_mov_8dst, "SALU"), C(c_s_and_b64, "SALU"), C(c_s_nop, "-"),
        { "c_mix5", c_mix5, BODY*5, "mixed: 2 fp64 + s_mov + cmp + cndmask per unit" },
        { "c_ieee_div_f64", c_ieee_div_f64, 8, "compiler's IEEE fp64 division (instruction sequence), per quotient" },
        { "c_ieee_sqrt_f64", c_ieee_sqrt_f64, 8, "compiler's IEEE fp64 sqrt (+1 add), per root" },
#undef C
    };
    Stamp *dStamps; double *dSink;
    const int maxWaves = simds*8;
    CHK(hipMalloc(&dStamps, sizeof(Stamp)*maxWaves));
    CHK(hipMalloc(&dSink, 8));
    std::vector<Stamp> h(maxWaves);
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (const Case &c : cases)
        for (int w : wps) {
            const int blocks = cus*w;                                  // 256-thread workgroups: one wave per SIMD of a CU, w workgroups per CU
            c.k<<<blocks, 256>>>(dStamps, dSink, 64);                  // warm-up (code fetch, clocks)
            CHK(hipEventRecord(e0));
            c.k<<<blocks, 256>>>(dStamps, dSink, ITERS);
            CHK(hipEventRecord(e1));
            CHK(hipEventSynchronize(e1));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
            const int waves = blocks*4;
            CHK(hipMemcpy(h.data(), dStamps, sizeof(Stamp)*waves, hipMemcpyDeviceToHost));
            std::vector<double> per(waves), mhz(waves);
            for (int i = 0; i < waves; ++i) {
                per[i] = (double) h[i].cycles/((double) ITERS*c.perIter)/w;   // SIMD cycles per instruction if exactly w waves share the SIMD
                mhz[i] = (double) h[i].cycles/((double) h[i].ticks100MHz/100.);
            }
            std::sort(per.begin(), per.end()); std::sort(mhz.begin(), mhz.end());
            const double clockMHz = mhz[waves/2];
            const double launchCycles = ms*1e-3*clockMHz*1e6*simds/((double) waves*ITERS*c.perIter);   // SIMD cycles per instruction over the whole launch
            printf("{\"case\": \"%s\", \"pmc_class\": \"%s\", \"waves_per_simd\": %d, \"cycles_per_inst_wave_median\": %.3f, \"cycles_per_inst_wave_min\": %.3f, "
                   "\"cycles_per_inst_launch\": %.3f, \"shader_clock_mhz\": %.0f, \"launch_ms\": %.4f, \"insts_per_wave\": %d}\n",
                   c.name, c.pmcClass, w, per[waves/2], per[0], launchCycles, clockMHz, ms, ITERS*c.perIter);
            fflush(stdout);
        }
    return 0;
}
