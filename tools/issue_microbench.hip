// issue_microbench.hip -- wave64 issue cost of the instructions the VEGAS sample loop is made of, measured on the
// box the bench runs on (MI355X, gfx950).  Built by __graft_entry__.build() (hipcc --offload-arch=gfx950) into mcintegration.jl_amd/lib/issue_microbench.
//
// Every test is ONE instruction form in inline asm, 8 independent destination registers, UNROLL copies per loop trip,
// run by W waves on every SIMD of the chip (grid = 256 CUs x W workgroups of 256 threads).  Each wave brackets its loop
// with s_memtime (shader-clock ticks) and the launch is timed with HIP events.  Two figures per row:
//     wall_ns_per_wave_inst_per_simd = launch time / (instructions per wave x W)
//         how long a SIMD (for ds_*: the CU's LDS pipe seen from one of its four SIMDs) is occupied per wave-instruction once
//         W waves compete for it -- THE issue cost (saturated from W = 4; it contains whatever clock the chip sustains)
//     cycles_per_wave_inst = mean wave ticks / (instructions per wave x W)
//         the same in shader-clock ticks of the waves themselves; exact at W = 1 and 2 (every wave resident from the start),
//         too low at W = 8 where not all waves of the grid overlap (ticks_per_wall_ns < clock shows it)
// Prints one JSON object per line.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef unsigned long long u64;
typedef unsigned int u32;

#define CHK(x)                                                                                  \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            fprintf(stderr, "%s: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                            \
        }                                                                                       \
    } while (0)

enum Op {
    OP_MOV_B32, OP_ADD_U32, OP_XOR_B32, OP_XOR_B32_E64, OP_XOR3_EMU, OP_BITOP3_B32, OP_BITOP3_VVV, OP_ADD3_U32, OP_LSHL_ADD_U32, OP_MAD_U64_U32, OP_MAD_U64_U32_SGPR, OP_MUL_LO_U32, OP_MUL_HI_U32, OP_LSHRREV_B64, OP_ALIGNBIT, OP_AND_OR_B32,
    OP_FMA_F64, OP_MUL_F64, OP_ADD_F64, OP_FRACT_F64, OP_CVT_I32_F64, OP_CVT_F64_I32, OP_FLOOR_F64, OP_LDEXP_F64, OP_RCP_F64, OP_CMP_F64, OP_CNDMASK,
    OP_FMA_F32, OP_PK_FMA_F32, OP_EXP_F32,
    OP_DS_READ_B128_RAND, OP_DS_READ_B64_RAND, OP_DS_READ_B64_SEQ, OP_DS_ADD_F64_RAND, OP_DS_ADD_F64_SAME, OP_DS_ADD_RTN_F64_RAND, OP_DS_ADD_F64_SEQ, OP_DS_ADD_U64_RAND, OP_DS_ADD_U32_RAND, OP_DS_ADD_F32_RAND, OP_DS_WRITE_B64_RAND,
    OP_DS_ADD_F64_COPY2, OP_DS_ADD_F64_COPY4, OP_DS_ADD_F64_COPY8, OP_DS_ADD_F64_COPY16,
    OP_GLOBAL_LOAD_B64_RAND_L2, OP_GLOBAL_LOAD_B128_RAND_L2, OP_GLOBAL_LOAD_B128_RAND_L1, OP_GLOBAL_LOAD_B128_UNALIGNED_L1,
    OP_COUNT
};
static const char *const kNames[OP_COUNT] = {
    "v_mov_b32", "v_add_u32", "v_xor_b32", "v_xor_b32_e64 (VOP3 encoding, 2 sources)", "v_xor_b32 x2 (3-input xor)", "v_bitop3_b32 (0x96 = xor3, one SGPR source)", "v_bitop3_b32 (three VGPR sources)", "v_add3_u32", "v_lshl_add_u32", "v_mad_u64_u32", "v_mad_u64_u32 (multiplier in an SGPR, as Philox has it)", "v_mul_lo_u32", "v_mul_hi_u32", "v_lshrrev_b64", "v_alignbit_b32", "v_and_or_b32",
    "v_fma_f64", "v_mul_f64", "v_add_f64", "v_fract_f64", "v_cvt_i32_f64", "v_cvt_f64_i32", "v_floor_f64", "v_ldexp_f64", "v_rcp_f64", "v_cmp_lt_f64", "v_cndmask_b32",
    "v_fma_f32", "v_pk_fma_f32", "v_exp_f32",
    "ds_read_b128 (random 16-B pairs, 999-bin table)", "ds_read_b64 (random, 999-bin table)", "ds_read_b64 (lane-consecutive)",
    "ds_add_f64 (random bins, 999-bin table)", "ds_add_f64 (one bin per wave)", "ds_add_rtn_f64 (random bins)", "ds_add_f64 (lane-consecutive bins: conflict-free)", "ds_add_u64 (random bins)", "ds_add_u32 (random bins)", "ds_add_f32 (random bins)", "ds_write_b64 (random bins)",
    "ds_add_f64 (random bins, 2 interleaved copies: lane l adds to copy l % 2)", "ds_add_f64 (random bins, 4 interleaved copies)", "ds_add_f64 (random bins, 8 interleaved copies)", "ds_add_f64 (random bins, 16 interleaved copies = 16 grids bin-major, lanes skewed)",
    "global_load_dwordx2 (random 8-B, 32 x 8 KB tables, L2-resident)", "global_load_dwordx4 (random 16-B, 32 x 16 KB tables, L2-resident)", "global_load_dwordx4 (random 16-B pairs of ONE 16 KB table: L1-resident)", "global_load_dwordx4 (random 8-B-aligned pairs of ONE 8 KB edge table: L1-resident)",
};

#define CLOB32 "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107"
#define CLOB64 "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115"
#define CLOB128 "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131"
#define UNROLL 8 // x 8 destinations = 64 instructions per loop trip

// REP8(X): X(0) .. X(7) -- one asm statement per destination register
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP> __global__ void __launch_bounds__(256) k_issue(u64 *ticks, int iters, const double *gtab, u32 seed) {
    extern __shared__ __attribute__((aligned(16))) double lds[]; // 2 x 999 doubles (pair table) | 999 doubles (histogram)
    const int tid = threadIdx.x, lane = tid & 63;
    constexpr int HC = OP == OP_DS_ADD_F64_COPY2 ? 2 : OP == OP_DS_ADD_F64_COPY4 ? 4 : OP == OP_DS_ADD_F64_COPY8 ? 8 : OP == OP_DS_ADD_F64_COPY16 ? 16 : 1;
    for (int i = tid; i < 2048 + 1024 * HC; i += 256) lds[i] = 1.0 + i;
    __syncthreads();
    // per-lane pseudo-random bins (LCG; what matters is that the lanes of a wave scatter like the map's draws do)
    u32 r = seed ^ (u32)(blockIdx.x * 256 + tid) * 2654435761u;
    u32 addr16[8], addr8[8], addrh[8];
    u64 gaddr8[8], gaddr16[8], gaddr1[8], gaddr1u[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        r = r * 1664525u + 1013904223u;
        const u32 bin = (u32)(((u64)(r >> 8) * 999ull) >> 24);
        addr16[j] = bin * 16u;            // (g, dx) pair j of a 999-bin PAIR_TABLE
        addr8[j] = bin * 8u;
        addrh[j] = 16384u + (HC == 1 ? bin * 8u : (bin * (u32)HC + ((u32)lane & (u32)(HC - 1))) * 8u); // histogram region (HC interleaved copies: mci_device.h hslot)
        gaddr8[j] = (u64)(gtab + (size_t)j * 4 * 1024 + bin);           // table j (of 32 x 8 KB)
        gaddr16[j] = (u64)(gtab + (size_t)j * 4 * 2048 + 2 * bin);      // table j (of 32 x 16 KB)
        gaddr1[j] = (u64)(gtab + 2 * bin);                              // ONE 16 KB table of (g, dx) pairs
        gaddr1u[j] = (u64)(gtab + bin);                                 // ONE 8 KB edge table: g[iy], g[iy+1] (8-byte aligned)
    }
    u32 a[8], b = r | 1u, c = r ^ 0x9E3779B9u;
    u64 q[8];
    double d[8], e = 1.0000001, f = 0.3;
    float s[8], sf = 1.0001f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef double d2 __attribute__((ext_vector_type(2)));
    f2 p[8], pf = {1.0001f, 0.9999f};
    d2 l2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        a[j] = r + j;
        q[j] = ((u64)r << 20) + j;
        d[j] = 1.0 + 0.001 * (lane + j);
        s[j] = 1.0f + 0.001f * (lane + j);
        p[j] = f2{s[j], s[j]};
        l2[j] = d2{0.0, 0.0};
    }
    const u32 seqaddr = (u32)lane * 8u;
    (void)seqaddr;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const u64 rt0 = __builtin_amdgcn_s_memrealtime();
    const u64 t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if constexpr (OP == OP_MOV_B32) {
#define X(j) asm volatile("v_mov_b32 v[100+" #j "], %0" : : "v"(b) : CLOB32);
                REP8(X)
#undef X
            } else if constexpr (OP == OP_ADD_U32) {
#define X(j) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[j]) : "v"(b));
                REP8(X)
#undef X
            } else if constexpr (OP == OP_XOR_B32) {
#define X(j) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a[j]) : "v"(b));
                REP8(X)
#undef X
            } else if constexpr (OP == OP_XOR_B32_E64) {
#define X(j) asm volatile("v_xor_b32_e64 %0, %1, %0" : "+v"(a[j]) : "v"(b));
                REP8(X)
#undef X
            } else if constexpr (OP == OP_BITOP3_VVV) {
#define X(j) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(a[j]) : "v"(b), "v"(c));
                REP8(X)
#undef X
            } else if constexpr (OP == OP_ADD3_U32) {
#define X(j) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c));
                REP8(X)
#undef X
            } else if constexpr (OP == OP_LSHL_ADD_U32) {
#define X(j) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a[j]) : "v"(b));
                REP8(X)
#undef X
            } else if constexpr (OP == OP_XOR3_EMU) { // a ^= b ^ c as the compiler has to emit it (no v_xor3_b32 on gfx950)
#define X(j) asm volatile("v_xor_b32 %0, %1, %0\n\tv_xor_b32 %0, %2, %0" : "+v"(a[j]) : "v"(b), "v"(c));
                REP8(X)
#undef X
            } else if constexpr (OP == OP_BITOP3_B32) { // what Philox uses: hi(product) ^ counter word ^ key word (SGPR)
#define X(j) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(a[j]) : "v"(b), "s"(seed));
                REP8(X)
#undef X
            } else if constexpr (OP == OP_MAD_U64_U32) { // Philox: (u64)M * c, hi and lo in one issue
#define X(j) asm volatile("v_mad_u64_u32 v[100+2*" #j ":101+2*" #j "], vcc, %0, %1, 0" : : "v"(b), "v"(a[j]) : "vcc", CLOB64);
                REP8(X)
#undef X
            } else if constexpr (OP == OP_MAD_U64_U32_SGPR) {
#define X(j) asm volatile("v_mad_u64_u32 v[100+2*" #j ":101+2*" #j "], vcc, %0, %1, 0" : : "s"(seed), "v"(a[j]) : "vcc", CLOB64);
                REP8(X)
#undef X
            } else if constexpr (OP == OP_MUL_LO_U32) {
#define X(j) asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(a[j]) : "v"(b));
                REP8(X)
#undef X
            } else if constexpr (OP == OP_MUL_HI_U32) {
#define X(j) asm volatile("v_mul_hi_u32 %0, %1, %0" : "+v"(a[j]) : "v"(b));
                REP8(X)
#undef X
            } else if constexpr (OP == OP_LSHRREV_B64) {
#define X(j) asm volatile("v_lshrrev_b64 v[100+2*" #j ":101+2*" #j "], 12, %0" : : "v"(q[j]) : CLOB64);
                REP8(X)
#undef X
            } else if constexpr (OP == OP_ALIGNBIT) {
#define X(j) asm volatile("v_alignbit_b32 %0, %1, %0, 12" : "+v"(a[j]) : "v"(b));
                REP8(X)
#undef X
            } else if constexpr (OP == OP_AND_OR_B32) {
#define X(j) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c));
                REP8(X)
#undef X
            } else if constexpr (OP == OP_FMA_F64) {
#define X(j) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[j]) : "v"(e), "v"(f));
                REP8(X)
#undef X
            } else if constexpr (OP == OP_MUL_F64) {
#define X(j) asm volatile("v_mul_f64 %0, %1, %0" : "+v"(d[j]) : "v"(e));
                REP8(X)
#undef X
            } else if constexpr (OP == OP_ADD_F64) {
#define X(j) asm volatile("v_add_f64 %0, %1, %0" : "+v"(d[j]) : "v"(f));
                REP8(X)
#undef X
            } else if constexpr (OP == OP_FRACT_F64) {
#define X(j) asm volatile("v_fract_f64 v[100+2*" #j ":101+2*" #j "], %0" : : "v"(d[j]) : CLOB64);
                REP8(X)
#undef X
            } else if constexpr (OP == OP_CVT_I32_F64) {
#define X(j) asm volatile("v_cvt_i32_f64 v[100+" #j "], %0" : : "v"(d[j]) : CLOB32);
                REP8(X)
#undef X
            } else if constexpr (OP == OP_CVT_F64_I32) {
#define X(j) asm volatile("v_cvt_f64_i32 v[100+2*" #j ":101+2*" #j "], %0" : : "v"(a[j]) : CLOB64);
                REP8(X)
#undef X
            } else if constexpr (OP == OP_FLOOR_F64) {
#define X(j) asm volatile("v_floor_f64 v[100+2*" #j ":101+2*" #j "], %0" : : "v"(d[j]) : CLOB64);
                REP8(X)
#undef X
            } else if constexpr (OP == OP_LDEXP_F64) {
#define X(j) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(d[j]) : "v"(b));
                REP8(X)
#undef X
            } else if constexpr (OP == OP_RCP_F64) {
#define X(j) asm volatile("v_rcp_f64 v[100+2*" #j ":101+2*" #j "], %0" : : "v"(d[j]) : CLOB64);
                REP8(X)
#undef X
            } else if constexpr (OP == OP_CMP_F64) {
#define X(j) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(d[j]), "v"(e) : "vcc");
                REP8(X)
#undef X
            } else if constexpr (OP == OP_CNDMASK) {
#define X(j) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[j]) : "v"(b) : "vcc");
                REP8(X)
#undef X
            } else if constexpr (OP == OP_FMA_F32) {
#define X(j) asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(s[j]) : "v"(sf));
                REP8(X)
#undef X
            } else if constexpr (OP == OP_PK_FMA_F32) {
#define X(j) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(p[j]) : "v"(pf));
                REP8(X)
#undef X
            } else if constexpr (OP == OP_EXP_F32) {
#define X(j) asm volatile("v_exp_f32 v[100+" #j "], %0" : : "v"(s[j]) : CLOB32);
                REP8(X)
#undef X
            } else if constexpr (OP == OP_DS_READ_B128_RAND) {
#define X(j) asm volatile("ds_read_b128 v[100+4*" #j ":103+4*" #j "], %0" : : "v"(addr16[j]) : "memory", CLOB128);
                REP8(X)
#undef X
            } else if constexpr (OP == OP_DS_READ_B64_RAND) {
#define X(j) asm volatile("ds_read_b64 v[100+2*" #j ":101+2*" #j "], %0" : : "v"(addr8[j]) : "memory", CLOB64);
                REP8(X)
#undef X
            } else if constexpr (OP == OP_DS_READ_B64_SEQ) {
#define X(j) asm volatile("ds_read_b64 v[100+2*" #j ":101+2*" #j "], %0 offset:" #j "*512" : : "v"(seqaddr) : "memory", CLOB64);
                REP8(X)
#undef X
            } else if constexpr (OP == OP_DS_ADD_F64_RAND || OP == OP_DS_ADD_F64_COPY2 || OP == OP_DS_ADD_F64_COPY4 || OP == OP_DS_ADD_F64_COPY8 || OP == OP_DS_ADD_F64_COPY16) {
#define X(j) asm volatile("ds_add_f64 %0, %1" : : "v"(addrh[j]), "v"(e) : "memory");
                REP8(X)
#undef X
            } else if constexpr (OP == OP_DS_ADD_F64_SAME) {
#define X(j) asm volatile("ds_add_f64 %0, %1 offset:" #j "*8" : : "v"(16384u), "v"(e) : "memory");
                REP8(X)
#undef X
            } else if constexpr (OP == OP_DS_ADD_RTN_F64_RAND) {
#define X(j) asm volatile("ds_add_rtn_f64 v[100+2*" #j ":101+2*" #j "], %0, %1" : : "v"(addrh[j]), "v"(e) : "memory", CLOB64);
                REP8(X)
#undef X
            } else if constexpr (OP == OP_DS_ADD_F64_SEQ) {
#define X(j) asm volatile("ds_add_f64 %0, %1 offset:" #j "*512" : : "v"(16384u + seqaddr), "v"(e) : "memory");
                REP8(X)
#undef X
            } else if constexpr (OP == OP_DS_ADD_U64_RAND) {
#define X(j) asm volatile("ds_add_u64 %0, %1" : : "v"(addrh[j]), "v"(q[j]) : "memory");
                REP8(X)
#undef X
            } else if constexpr (OP == OP_DS_ADD_U32_RAND) {
#define X(j) asm volatile("ds_add_u32 %0, %1" : : "v"(addrh[j]), "v"(b) : "memory");
                REP8(X)
#undef X
            } else if constexpr (OP == OP_DS_ADD_F32_RAND) {
#define X(j) asm volatile("ds_add_f32 %0, %1" : : "v"(addrh[j]), "v"(sf) : "memory");
                REP8(X)
#undef X
            } else if constexpr (OP == OP_DS_WRITE_B64_RAND) {
#define X(j) asm volatile("ds_write_b64 %0, %1" : : "v"(addrh[j]), "v"(e) : "memory");
                REP8(X)
#undef X
            } else if constexpr (OP == OP_GLOBAL_LOAD_B64_RAND_L2) {
#define X(j) asm volatile("global_load_dwordx2 v[100+2*" #j ":101+2*" #j "], %0, off" : : "v"(gaddr8[j]) : "memory", CLOB64);
                REP8(X)
#undef X
            } else if constexpr (OP == OP_GLOBAL_LOAD_B128_RAND_L1) {
#define X(j) asm volatile("global_load_dwordx4 v[100+4*" #j ":103+4*" #j "], %0, off" : : "v"(gaddr1[j]) : "memory", CLOB128);
                REP8(X)
#undef X
            } else if constexpr (OP == OP_GLOBAL_LOAD_B128_UNALIGNED_L1) {
#define X(j) asm volatile("global_load_dwordx4 v[100+4*" #j ":103+4*" #j "], %0, off" : : "v"(gaddr1u[j]) : "memory", CLOB128);
                REP8(X)
#undef X
            } else if constexpr (OP == OP_GLOBAL_LOAD_B128_RAND_L2) {
#define X(j) asm volatile("global_load_dwordx4 v[100+4*" #j ":103+4*" #j "], %0, off" : : "v"(gaddr16[j]) : "memory", CLOB128);
                REP8(X)
#undef X
            }
        }
        if constexpr (OP >= OP_DS_READ_B128_RAND) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const u64 t1 = __builtin_readcyclecounter();
    const u64 rt1 = __builtin_amdgcn_s_memrealtime();
    // keep every destination alive
    u32 sink = 0;
    double dsink = 0.0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        sink ^= a[j] ^ (u32)q[j] ^ (u32)(q[j] >> 32) ^ __float_as_uint(s[j]) ^ __float_as_uint(p[j].x) ^ __float_as_uint(p[j].y);
        dsink += d[j] + l2[j].x + l2[j].y;
    }
    if (sink == 0x12345678u && dsink == 1.2345) ticks[0] = 0;
    if (lane == 0) {
        ticks[(size_t)blockIdx.x * 4 + (tid >> 6)] = t1 - t0;
        ticks[((size_t)gridDim.x + blockIdx.x) * 4 + (tid >> 6)] = rt1 - rt0; // reference-clock ticks (100 MHz) of the same stretch
    }
}

// ---- the Philox4x32-10 block exactly as mci_device.h writes it (compiler-scheduled), per-call cost ----
struct u32x4 { u32 x, y, z, w; };
__device__ __forceinline__ u32x4 philox(u32 c0, u32 c1, u32 c2, u32 c3, u32 k0, u32 k1, int rounds) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        if (r < rounds) {
            const u64 p0 = (u64)0xD2511F53u * c0;
            const u64 p1 = (u64)0xCD9E8D57u * c2;
            const u32 n0 = __builtin_amdgcn_bitop3_b32((u32)(p1 >> 32), c1, k0, 0x96);
            const u32 n2 = __builtin_amdgcn_bitop3_b32((u32)(p0 >> 32), c3, k1, 0x96);
            c1 = (u32)p1;
            c3 = (u32)p0;
            c0 = n0;
            c2 = n2;
            k0 += 0x9E3779B9u;
            k1 += 0xBB67AE85u;
        }
    }
    return {c0, c1, c2, c3};
}
template <int ROUNDS> __global__ void __launch_bounds__(256) k_philox(u64 *ticks, int iters, u32 seed) {
    const int tid = threadIdx.x, lane = tid & 63;
    u32 c0 = blockIdx.x * 256 + tid, c1 = seed, acc = 0;
    const u64 t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) { // 8 independent calls per trip: the sample loop's 8 chunks of a 16-D draw
            const u32x4 r = philox(c0, c1, (u32)u, (u32)it, seed, 0x1234567u, ROUNDS);
            acc ^= r.x ^ r.y ^ r.z ^ r.w;
        }
        c0 += 65536u;
    }
    const u64 t1 = __builtin_readcyclecounter();
    if (acc == 0x12345678u) ticks[0] = 0;
    if (lane == 0) {
        ticks[(size_t)blockIdx.x * 4 + (tid >> 6)] = t1 - t0;
        ticks[((size_t)gridDim.x + blockIdx.x) * 4 + (tid >> 6)] = 0;
    }
}

static double g_wall_khz = 100000.0; // hipDeviceAttributeWallClockRate
struct Res { double cyc_per_inst, ns_per_inst, clock_ghz, slope_ns, sclk_mhz; };

// launch(nblk, iters).  ns_per_inst = wall time of one launch / wave-instructions per SIMD (includes launch + tail: an upper
// bound); slope_ns = (wall(2 * iters) - wall(iters)) / the extra wave-instructions: the fixed part cancels -- the issue cost.
template <class F> static Res run(F launch, int W, double insts_per_wave_per_iter, int iters, u64 *d_ticks, int ncu) {
    const int nblk = ncu * W;
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    auto timed = [&](int it) {
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            CHK(hipEventRecord(e0));
            launch(nblk, it);
            CHK(hipEventRecord(e1));
            CHK(hipDeviceSynchronize());
            float ms = 0.f;
            CHK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        return best;
    };
    launch(nblk, iters); // warm-up
    CHK(hipDeviceSynchronize());
    const float ms2 = timed(2 * iters);
    const float ms = timed(iters); // (last: the tick counters below are this launch's)
    std::vector<u64> t((size_t)nblk * 4 * 2);
    CHK(hipMemcpy(t.data(), d_ticks, t.size() * sizeof(u64), hipMemcpyDeviceToHost));
    double mean = 0.0, rmean = 0.0;
    for (size_t k = 0; k < t.size() / 2; ++k) {
        mean += (double)t[k];
        rmean += (double)t[t.size() / 2 + k];
    }
    mean /= (double)(t.size() / 2);
    rmean /= (double)(t.size() / 2);
    CHK(hipEventDestroy(e0));
    CHK(hipEventDestroy(e1));
    const double n = insts_per_wave_per_iter * iters;
    Res r;
    r.cyc_per_inst = mean / (n * W);
    r.ns_per_inst = (double)ms * 1e6 / (n * W);
    r.clock_ghz = mean / ((double)ms * 1e6);                // ticks per ns if a wave spans the whole launch
    r.slope_ns = ((double)ms2 - (double)ms) * 1e6 / (n * W);
    // shader clock during the measured stretch: s_memtime ticks (shader cycles) per s_memrealtime tick (constant rate: g_wall_khz)
    r.sclk_mhz = rmean > 0.0 ? mean / rmean * g_wall_khz * 1e-3 : 0.0;
    return r;
}

static bool g_roofline_only = false;
static bool in_roofline_set(int op) { // the forms the sample loop's mix is priced with (mcintegration.jl_amd/isa_mix.py COST_KEY)
    switch (op) {
    case OP_XOR_B32: case OP_ALIGNBIT: case OP_BITOP3_B32: case OP_MAD_U64_U32: case OP_MUL_LO_U32: case OP_LSHRREV_B64: case OP_FMA_F64: case OP_MUL_F64:
    case OP_ADD_F64: case OP_FRACT_F64: case OP_CVT_I32_F64: case OP_RCP_F64: case OP_CMP_F64: case OP_LDEXP_F64: case OP_EXP_F32:
    case OP_DS_READ_B128_RAND: case OP_DS_READ_B64_RAND: case OP_DS_ADD_F64_RAND:
    case OP_DS_ADD_F64_COPY2: case OP_DS_ADD_F64_COPY4: case OP_DS_ADD_F64_COPY8: case OP_DS_ADD_F64_COPY16:
        return true;
    default:
        return false;
    }
}

template <int OP> static void bench_op(u64 *d_ticks, const double *d_gtab, int ncu, int iters) {
    if (g_roofline_only && !in_roofline_set(OP)) return;
    for (int W : {1, 2, 4, 8}) {
        if (g_roofline_only && W < 4) continue;
        constexpr int HC = OP == OP_DS_ADD_F64_COPY2 ? 2 : OP == OP_DS_ADD_F64_COPY4 ? 4 : OP == OP_DS_ADD_F64_COPY8 ? 8 : OP == OP_DS_ADD_F64_COPY16 ? 16 : 1;
        constexpr size_t lds = (2048 + 1024 * HC) * sizeof(double);
        if (lds > 64 * 1024) CHK(hipFuncSetAttribute((const void *)k_issue<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        auto launch = [&](int nblk, int it) { hipLaunchKernelGGL(k_issue<OP>, dim3(nblk), dim3(256), lds, 0, d_ticks, it, d_gtab, 12345u); };
        const double n = (double)UNROLL * 8 * (OP == OP_XOR3_EMU ? 2 : 1);
        const Res r = run(launch, W, n, iters, d_ticks, ncu);
        printf("{\"op\": \"%s\", \"waves_per_simd\": %d, \"cycles_per_wave_inst\": %.3f, \"wall_ns_per_wave_inst_per_simd\": %.4f, \"slope_ns_per_wave_inst_per_simd\": %.4f, \"ticks_per_wall_ns\": %.3f, \"sclk_mhz\": %.1f}\n",
               kNames[OP], W, r.cyc_per_inst, r.ns_per_inst, r.slope_ns, r.clock_ghz, r.sclk_mhz);
        fflush(stdout);
    }
}

template <int OP> static void bench_all(u64 *d_ticks, const double *d_gtab, int ncu, int iters) {
    if constexpr (OP < OP_COUNT) {
        bench_op<OP>(d_ticks, d_gtab, ncu, iters);
        bench_all<OP + 1>(d_ticks, d_gtab, ncu, iters);
    }
}

template <int ROUNDS> static void bench_philox(u64 *d_ticks, int ncu, int iters) {
    for (int W : {1, 2, 4, 8}) {
        auto launch = [&](int nblk, int it) { hipLaunchKernelGGL(k_philox<ROUNDS>, dim3(nblk), dim3(256), 0, 0, d_ticks, it, 777u); };
        const Res r = run(launch, W, 8.0, iters, d_ticks, ncu);
        printf("{\"op\": \"philox4x32-%d call (compiler-scheduled, 2 v_mad_u64_u32 + 2 v_bitop3_b32 per round)\", \"waves_per_simd\": %d, \"cycles_per_wave_inst\": %.2f, "
               "\"wall_ns_per_wave_inst_per_simd\": %.4f, \"slope_ns_per_wave_inst_per_simd\": %.4f, \"ticks_per_wall_ns\": %.3f}\n", ROUNDS, W, r.cyc_per_inst, r.ns_per_inst, r.slope_ns, r.clock_ghz);
        fflush(stdout);
    }
}

int main(int argc, char **argv) {
    // usage: issue_microbench [iters] [roofline]   ("roofline": only the forms bench.py prices the sample loop with, 4 and 8 waves/SIMD)
    int iters = argc > 1 ? atoi(argv[1]) : 2000;
    g_roofline_only = argc > 2 && strcmp(argv[2], "roofline") == 0;
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    printf("{\"device\": \"%s\", \"arch\": \"%s\", \"cus\": %d, \"clock_mhz\": %d, \"iters\": %d, \"insts_per_trip\": %d}\n", prop.name, prop.gcnArchName, ncu,
           prop.clockRate / 1000, iters, UNROLL * 8);
    u64 *d_ticks;
    CHK(hipMalloc((void **)&d_ticks, (size_t)ncu * 8 * 4 * 2 * sizeof(u64)));
    {
        int khz = 0;
        if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0) == hipSuccess && khz > 0) g_wall_khz = (double)khz;
    }
    double *d_gtab;
    CHK(hipMalloc((void **)&d_gtab, 32 * 2048 * sizeof(double)));
    CHK(hipMemset(d_gtab, 0, 32 * 2048 * sizeof(double)));
    bench_all<0>(d_ticks, d_gtab, ncu, iters);
    if (!g_roofline_only) {
        bench_philox<10>(d_ticks, ncu, iters / 4 > 0 ? iters / 4 : 1);
        bench_philox<7>(d_ticks, ncu, iters / 4 > 0 ? iters / 4 : 1);
    }
    return 0;
}
