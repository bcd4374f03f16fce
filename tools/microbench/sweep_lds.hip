// sweep_lds.hip -- more bytes in flight for the wide sweep WITHOUT spending registers on them?
//
// k_sweepw<24> (product) streams a tile of rows through a thread that keeps its 24 prow pairs in
// 96 VGPRs: two register sets of four rows, i.e. ONE step of four rows (4 KB per wave) travels while
// one is computed.  rocprofv3 + PMC (round 4): 121.5 us per pass at config 3 = HBM time (67 us) plus
// f64 time (41 us) almost serially; every way of buying a deeper prefetch with registers lost to
// three resident waves (DESIGN_experiments.md R4.7).  gfx950 can load global memory straight into
// LDS (global_load_lds_dwordx4: 1 KB per wave-instruction, no VGPR involved): here the rows of the
// next D steps travel into a per-wave LDS ring, a step's rows are read back with ds_read_b128 when
// their turn comes, waited for with a COUNTED s_waitcnt vmcnt so that the younger prefetches and the
// stores stay in flight.  No barrier anywhere: a wave only ever reads what it loaded itself.
//
//   variant R   the product's structure: register prefetch of one step        (baseline)
//   variant L   LDS ring, D steps ahead (D = 2, 3), U = 4 rows per step
// Same links (four per asm statement, operands in fixed SGPRs), same roundings; results are compared
// bit for bit between the variants.
//
// usage: sweep_lds [rows ld]     default 4097 x 8208 (config 3, compact); try 32769 8208 (one shard)
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o sweep_lds sweep_lds.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <vector>

typedef double vec2d __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

#define MI_W4_ROW(S, X, Y, PX, PY)                                                                   \
    "v_mul_f64 %[t0], " S ", %[" PX "]\n\t" "v_mul_f64 %[t1], " S ", %[" PY "]\n\t"                   \
    "v_add_f64 %[" X "], %[" X "], -%[t0]\n\t" "v_add_f64 %[" Y "], %[" Y "], -%[t1]\n\t"
#define MI_W4_LINK(S0, S1, S2, S3, PX, PY)                                                          \
    MI_W4_ROW(S0, "x0", "y0", PX, PY) MI_W4_ROW(S1, "x1", "y1", PX, PY)                             \
    MI_W4_ROW(S2, "x2", "y2", PX, PY) MI_W4_ROW(S3, "x3", "y3", PX, PY)
__device__ __forceinline__ void links4(vec2d (&cur)[4], const vec2d pa, const vec2d pb, const vec2d pc,
                                       const vec2d pd, const double *base, unsigned o1, unsigned o2, unsigned o3)
{
    double t0, t1;
    asm volatile(
        "s_load_dwordx8 s[36:43], %[base], 0x0\n\t"
        "s_load_dwordx8 s[44:51], %[base], %[o1]\n\t"
        "s_load_dwordx8 s[52:59], %[base], %[o2]\n\t"
        "s_load_dwordx8 s[60:67], %[base], %[o3]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        MI_W4_LINK("s[36:37]", "s[38:39]", "s[40:41]", "s[42:43]", "pax", "pay")
        MI_W4_LINK("s[44:45]", "s[46:47]", "s[48:49]", "s[50:51]", "pbx", "pby")
        MI_W4_LINK("s[52:53]", "s[54:55]", "s[56:57]", "s[58:59]", "pcx", "pcy")
        MI_W4_LINK("s[60:61]", "s[62:63]", "s[64:65]", "s[66:67]", "pdx", "pdy")
        : [x0] "+v"(cur[0].x), [y0] "+v"(cur[0].y), [x1] "+v"(cur[1].x), [y1] "+v"(cur[1].y),
          [x2] "+v"(cur[2].x), [y2] "+v"(cur[2].y), [x3] "+v"(cur[3].x), [y3] "+v"(cur[3].y),
          [t0] "=&v"(t0), [t1] "=&v"(t1)
        : [pax] "v"(pa.x), [pay] "v"(pa.y), [pbx] "v"(pb.x), [pby] "v"(pb.y),
          [pcx] "v"(pc.x), [pcy] "v"(pc.y), [pdx] "v"(pd.x), [pdy] "v"(pd.y),
          [base] "s"(base), [o1] "s"(o1), [o2] "s"(o2), [o3] "s"(o3)
        : "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51",
          "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67");
}

// the same four links with EIGHT temporaries: the eight products of a link first, then the eight
// differences -- a dependent instruction follows its producer eight instructions later instead of two
#define MI_T8_LINK(S0, S1, S2, S3, PX, PY)                                                          \
    "v_mul_f64 %[t0], " S0 ", %[" PX "]\n\t" "v_mul_f64 %[t1], " S0 ", %[" PY "]\n\t"               \
    "v_mul_f64 %[t2], " S1 ", %[" PX "]\n\t" "v_mul_f64 %[t3], " S1 ", %[" PY "]\n\t"               \
    "v_mul_f64 %[t4], " S2 ", %[" PX "]\n\t" "v_mul_f64 %[t5], " S2 ", %[" PY "]\n\t"               \
    "v_mul_f64 %[t6], " S3 ", %[" PX "]\n\t" "v_mul_f64 %[t7], " S3 ", %[" PY "]\n\t"               \
    "v_add_f64 %[x0], %[x0], -%[t0]\n\t" "v_add_f64 %[y0], %[y0], -%[t1]\n\t"                       \
    "v_add_f64 %[x1], %[x1], -%[t2]\n\t" "v_add_f64 %[y1], %[y1], -%[t3]\n\t"                       \
    "v_add_f64 %[x2], %[x2], -%[t4]\n\t" "v_add_f64 %[y2], %[y2], -%[t5]\n\t"                       \
    "v_add_f64 %[x3], %[x3], -%[t6]\n\t" "v_add_f64 %[y3], %[y3], -%[t7]\n\t"
__device__ __forceinline__ void links4t8(vec2d (&cur)[4], const vec2d pa, const vec2d pb, const vec2d pc,
                                         const vec2d pd, const double *base, unsigned o1, unsigned o2, unsigned o3)
{
    double t0, t1, t2, t3, t4, t5, t6, t7;
    asm volatile(
        "s_load_dwordx8 s[36:43], %[base], 0x0\n\t"
        "s_load_dwordx8 s[44:51], %[base], %[o1]\n\t"
        "s_load_dwordx8 s[52:59], %[base], %[o2]\n\t"
        "s_load_dwordx8 s[60:67], %[base], %[o3]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        MI_T8_LINK("s[36:37]", "s[38:39]", "s[40:41]", "s[42:43]", "pax", "pay")
        MI_T8_LINK("s[44:45]", "s[46:47]", "s[48:49]", "s[50:51]", "pbx", "pby")
        MI_T8_LINK("s[52:53]", "s[54:55]", "s[56:57]", "s[58:59]", "pcx", "pcy")
        MI_T8_LINK("s[60:61]", "s[62:63]", "s[64:65]", "s[66:67]", "pdx", "pdy")
        : [x0] "+v"(cur[0].x), [y0] "+v"(cur[0].y), [x1] "+v"(cur[1].x), [y1] "+v"(cur[1].y),
          [x2] "+v"(cur[2].x), [y2] "+v"(cur[2].y), [x3] "+v"(cur[3].x), [y3] "+v"(cur[3].y),
          [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [t4] "=&v"(t4), [t5] "=&v"(t5), [t6] "=&v"(t6), [t7] "=&v"(t7)
        : [pax] "v"(pa.x), [pay] "v"(pa.y), [pbx] "v"(pb.x), [pby] "v"(pb.y),
          [pcx] "v"(pc.x), [pcy] "v"(pc.y), [pdx] "v"(pd.x), [pdy] "v"(pd.y),
          [base] "s"(base), [o1] "s"(o1), [o2] "s"(o2), [o3] "s"(o3)
        : "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51",
          "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67");
}

// EIGHT links per statement: the second chunk of four pivots is requested as soon as the first has
// arrived and travels while the first is applied -- one exposed scalar-load wait per eight links
// instead of two (30 operands: the most one asm statement takes)
__device__ __forceinline__ void links8(vec2d (&cur)[4], const vec2d pa, const vec2d pb, const vec2d pc, const vec2d pd,
                                       const vec2d pe, const vec2d pf, const vec2d pg, const vec2d ph,
                                       const double *base, const double *base2, unsigned o1, unsigned o2)
{
    double t0, t1;
    asm volatile(
        "s_load_dwordx8 s[36:43], %[base], 0x0\n\t"
        "s_load_dwordx8 s[44:51], %[base], %[o1]\n\t"
        "s_load_dwordx8 s[52:59], %[base], %[o2]\n\t"
        "s_add_u32 s100, %[o1], %[o2]\n\t"
        "s_load_dwordx8 s[60:67], %[base], s100\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "s_load_dwordx8 s[68:75], %[base2], 0x0\n\t"
        "s_load_dwordx8 s[76:83], %[base2], %[o1]\n\t"
        "s_load_dwordx8 s[84:91], %[base2], %[o2]\n\t"
        "s_load_dwordx8 s[92:99], %[base2], s100\n\t"
        MI_W4_LINK("s[36:37]", "s[38:39]", "s[40:41]", "s[42:43]", "pax", "pay")
        MI_W4_LINK("s[44:45]", "s[46:47]", "s[48:49]", "s[50:51]", "pbx", "pby")
        MI_W4_LINK("s[52:53]", "s[54:55]", "s[56:57]", "s[58:59]", "pcx", "pcy")
        MI_W4_LINK("s[60:61]", "s[62:63]", "s[64:65]", "s[66:67]", "pdx", "pdy")
        "s_waitcnt lgkmcnt(0)\n\t"
        MI_W4_LINK("s[68:69]", "s[70:71]", "s[72:73]", "s[74:75]", "pex", "pey")
        MI_W4_LINK("s[76:77]", "s[78:79]", "s[80:81]", "s[82:83]", "pfx", "pfy")
        MI_W4_LINK("s[84:85]", "s[86:87]", "s[88:89]", "s[90:91]", "pgx", "pgy")
        MI_W4_LINK("s[92:93]", "s[94:95]", "s[96:97]", "s[98:99]", "phx", "phy")
        : [x0] "+v"(cur[0].x), [y0] "+v"(cur[0].y), [x1] "+v"(cur[1].x), [y1] "+v"(cur[1].y),
          [x2] "+v"(cur[2].x), [y2] "+v"(cur[2].y), [x3] "+v"(cur[3].x), [y3] "+v"(cur[3].y),
          [t0] "=&v"(t0), [t1] "=&v"(t1)
        : [pax] "v"(pa.x), [pay] "v"(pa.y), [pbx] "v"(pb.x), [pby] "v"(pb.y),
          [pcx] "v"(pc.x), [pcy] "v"(pc.y), [pdx] "v"(pd.x), [pdy] "v"(pd.y),
          [pex] "v"(pe.x), [pey] "v"(pe.y), [pfx] "v"(pf.x), [pfy] "v"(pf.y),
          [pgx] "v"(pg.x), [pgy] "v"(pg.y), [phx] "v"(ph.x), [phy] "v"(ph.y),
          [base] "s"(base), [base2] "s"(base2), [o1] "s"(o1), [o2] "s"(o2)
        : "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51",
          "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67",
          "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83",
          "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99", "s100", "scc");
}

// The CONTINUOUS chain (round 5, second half): a statement never waits for a load it has just issued.
// On entry the first chunk (pivots i0 .. i0+3, s[68:99] -- the HIGHEST registers: the compiler knows them
// only as clobbers and takes the lowest free ones for what it needs between two statements) is already on its way -- requested by the
// statement before this one (or the prologue); it asks for the second chunk, applies the first, and at
// its end asks for the NEXT statement's first chunk (nbase: the next pivots of these rows, or pivots
// 0 .. 3 of the next step's rows), which travels while the second chunk is applied.
__device__ __forceinline__ void links8c(vec2d (&cur)[4], const vec2d pa, const vec2d pb, const vec2d pc, const vec2d pd,
                                        const vec2d pe, const vec2d pf, const vec2d pg, const vec2d ph,
                                        const double *base2, const double *nbase, unsigned o1, unsigned o2)
{
    double t0, t1;
    asm volatile(
        "s_add_u32 s34, %[o1], %[o2]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "s_load_dwordx8 s[36:43], %[base2], 0x0\n\t"
        "s_load_dwordx8 s[44:51], %[base2], %[o1]\n\t"
        "s_load_dwordx8 s[52:59], %[base2], %[o2]\n\t"
        "s_load_dwordx8 s[60:67], %[base2], s34\n\t"
        MI_W4_LINK("s[68:69]", "s[70:71]", "s[72:73]", "s[74:75]", "pax", "pay")
        MI_W4_LINK("s[76:77]", "s[78:79]", "s[80:81]", "s[82:83]", "pbx", "pby")
        MI_W4_LINK("s[84:85]", "s[86:87]", "s[88:89]", "s[90:91]", "pcx", "pcy")
        MI_W4_LINK("s[92:93]", "s[94:95]", "s[96:97]", "s[98:99]", "pdx", "pdy")
        "s_waitcnt lgkmcnt(0)\n\t"
        "s_load_dwordx8 s[68:75], %[nbase], 0x0\n\t"
        "s_load_dwordx8 s[76:83], %[nbase], %[o1]\n\t"
        "s_load_dwordx8 s[84:91], %[nbase], %[o2]\n\t"
        "s_load_dwordx8 s[92:99], %[nbase], s34\n\t"
        MI_W4_LINK("s[36:37]", "s[38:39]", "s[40:41]", "s[42:43]", "pex", "pey")
        MI_W4_LINK("s[44:45]", "s[46:47]", "s[48:49]", "s[50:51]", "pfx", "pfy")
        MI_W4_LINK("s[52:53]", "s[54:55]", "s[56:57]", "s[58:59]", "pgx", "pgy")
        MI_W4_LINK("s[60:61]", "s[62:63]", "s[64:65]", "s[66:67]", "phx", "phy")
        : [x0] "+v"(cur[0].x), [y0] "+v"(cur[0].y), [x1] "+v"(cur[1].x), [y1] "+v"(cur[1].y),
          [x2] "+v"(cur[2].x), [y2] "+v"(cur[2].y), [x3] "+v"(cur[3].x), [y3] "+v"(cur[3].y),
          [t0] "=&v"(t0), [t1] "=&v"(t1)
        : [pax] "v"(pa.x), [pay] "v"(pa.y), [pbx] "v"(pb.x), [pby] "v"(pb.y),
          [pcx] "v"(pc.x), [pcy] "v"(pc.y), [pdx] "v"(pd.x), [pdy] "v"(pd.y),
          [pex] "v"(pe.x), [pey] "v"(pe.y), [pfx] "v"(pf.x), [pfy] "v"(pf.y),
          [pgx] "v"(pg.x), [pgy] "v"(pg.y), [phx] "v"(ph.x), [phy] "v"(ph.y),
          [base2] "s"(base2), [nbase] "s"(nbase), [o1] "s"(o1), [o2] "s"(o2)
        : "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51",
          "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67",
          "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83",
          "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99", "s34", "scc", "memory");
}
// the chain's first request (a tile's prologue): pivots 0 .. 3 of the first step's rows
__device__ __forceinline__ void links_first(const double *base, unsigned o1, unsigned o2)
{
    asm volatile(
        "s_add_u32 s34, %[o1], %[o2]\n\t"
        "s_load_dwordx8 s[68:75], %[base], 0x0\n\t"
        "s_load_dwordx8 s[76:83], %[base], %[o1]\n\t"
        "s_load_dwordx8 s[84:91], %[base], %[o2]\n\t"
        "s_load_dwordx8 s[92:99], %[base], s34\n\t"
        :: [base] "s"(base), [o1] "s"(o1), [o2] "s"(o2)
        : "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83",
          "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99", "s34", "scc", "memory");
}

// ---- variant R: the product's structure (k_sweepw without masks / slots)
template <int K, bool NT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 8)))
void k_reg(double *M, const double *prow, const double *col, int64_t ld, int64_t rows, int64_t col_stride, int tr, int strip_pairs)
{
    constexpr int U = 4;
    const int64_t ldv = ld >> 1;
    const int64_t pair = (int64_t)blockIdx.x * strip_pairs + threadIdx.x;
    if (!((int)threadIdx.x < strip_pairs && pair < ldv)) return;
    const int64_t r0 = (int64_t)blockIdx.y * tr;
    const int64_t r1 = (r0 + tr < rows) ? r0 + tr : rows;
    char *const Mb = reinterpret_cast<char *>(M);
    const int64_t row_bytes = ld * 8;
    const unsigned lane_off = (unsigned)pair * 16u;
    auto lane = [&]() -> unsigned { unsigned o = lane_off; asm volatile("" : "+v"(o)); return o; };
    auto ld2 = [&](int64_t r) -> vec2d {
        const vec2d *q = reinterpret_cast<const vec2d *>(Mb + r * row_bytes + lane());
        if constexpr (NT) return __builtin_nontemporal_load(q);
        else              return *q;
    };
    auto st2 = [&](int64_t r, vec2d v) {
        vec2d *q = reinterpret_cast<vec2d *>(Mb + r * row_bytes + lane());
        if constexpr (NT) __builtin_nontemporal_store(v, q);
        else              *q = v;
    };
    vec2d xa[U], xb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { xa[u].x = 0.0; xa[u].y = 0.0; xb[u] = xa[u]; if (r0 + u < r1) xa[u] = ld2(r0 + u); }
    vec2d p[K];
#pragma unroll
    for (int i = 0; i < K; ++i) p[i] = reinterpret_cast<const vec2d *>(prow)[(int64_t)i * ldv + pair];
    const unsigned o1 = (unsigned)(col_stride * 8), o2 = 2u * o1, o3 = 3u * o1;
    const int64_t group_stride = (int64_t)4 * col_stride;
    auto step = [&](vec2d (&cur)[U], vec2d (&nxt)[U], const int64_t r) {
#pragma unroll
        for (int u = 0; u < U; ++u) if (r + U + u < r1) nxt[u] = ld2(r + U + u);
        const double *cb = col + r;
#pragma unroll
        for (int i0 = 0; i0 < K; i0 += 4)
            links4(cur, p[i0], p[i0 + 1], p[i0 + 2], p[i0 + 3], cb + (int64_t)(i0 / 4) * group_stride, o1, o2, o3);
#pragma unroll
        for (int u = 0; u < U; ++u) if (r + u < r1) st2(r + u, cur[u]);
    };
    for (int64_t r = r0; r < r1; r += 2 * U) {
        step(xa, xb, r);
        if (r + U < r1) step(xb, xa, r + U);
    }
}

// ---- variant L: the next D steps' rows travel into a per-wave LDS ring
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

template <int K, int D, bool NT, int MODE = 0, int WMAX = 8, int L8 = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, WMAX)))
void k_lds(double *M, const double *prow, const double *col, int64_t ld, int64_t rows, int64_t col_stride, int tr, int strip_pairs)
{
    constexpr int U = 4;
    extern __shared__ __attribute__((aligned(16))) char ring_all[];           // 4 waves x D slots x 4 rows x 1 KB
    const int64_t ldv = ld >> 1;
    const int64_t pair = (int64_t)blockIdx.x * strip_pairs + threadIdx.x;
    if (!((int)threadIdx.x < strip_pairs && pair < ldv)) return;
    const int64_t r0 = (int64_t)blockIdx.y * tr;
    const int64_t r1 = (r0 + tr < rows) ? r0 + tr : rows;
    const int nsteps = (int)((r1 - r0 + U - 1) / U);
    char *const Mb = reinterpret_cast<char *>(M);
    const int64_t row_bytes = ld * 8;
    const unsigned lane_off = (unsigned)pair * 16u;
    auto lane = [&]() -> unsigned { unsigned o = lane_off; asm volatile("" : "+v"(o)); return o; };
    const int wave = (int)(threadIdx.x >> 6);
    char *const ring = ring_all + wave * (D * U * 1024);                      // wave-uniform
    const unsigned my_lds = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char *)ring) + (threadIdx.x & 63u) * 16u;   // this lane's 16 bytes of a row slot
    // rows of step s -> slot s % D (rows past the tile's end are clamped: loaded, never stored)
    auto issue = [&](int s) {
        const int slot = s % D;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int64_t r = r0 + (int64_t)s * U + u;
            if (r > r1 - 1) r = r1 - 1;
            if constexpr (MODE == 2) continue;
            const char *g = Mb + r * row_bytes + lane();
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                             (__attribute__((address_space(3))) void *)(ring + (slot * U + u) * 1024),
                                             16, 0, NT ? 2 : 0);
        }
    };
    auto fetch = [&](vec2d (&cur)[U], int s) {
        const unsigned a = my_lds + (unsigned)((s % D) * U * 1024);
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\t"
                     "ds_read_b128 %2, %4 offset:2048\n\tds_read_b128 %3, %4 offset:3072\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(cur[0]), "=&v"(cur[1]), "=&v"(cur[2]), "=&v"(cur[3]) : "v"(a) : "memory");
    };
    auto st2 = [&](int64_t r, vec2d v) {
        vec2d *q = reinterpret_cast<vec2d *>(Mb + r * row_bytes + lane());
        if constexpr (NT) __builtin_nontemporal_store(v, q);
        else              *q = v;
    };
    // (the prow pairs are ordinary loads: requested BEFORE the ring's first rows, so that the compiler's
    // wait for them is a counted one that leaves the ring's loads in flight)
    vec2d p[K];
#pragma unroll
    for (int i = 0; i < K; ++i) p[i] = reinterpret_cast<const vec2d *>(prow)[(int64_t)i * ldv + pair];
    asm volatile("" ::: "memory");
#pragma unroll
    for (int s = 0; s < D; ++s) if (s < nsteps) issue(s);
    const unsigned o1 = (unsigned)(col_stride * 8), o2 = 2u * o1, o3 = 3u * o1;
    const int64_t group_stride = (int64_t)4 * col_stride;
    // the prow loads are ordinary loads issued AFTER the first D steps' row loads: make sure they are
    // here (the compiler waits for them at first use -- with vmcnt(0), which also drains the ring's
    // first loads; once, in the prologue)
    vec2d cur[U];
    for (int s = 0; s < nsteps; ++s) {
        // in flight behind the rows of step s: the rows of steps s+1 .. s+D-1 and the stores of the D
        // steps before this one (4 instructions each); at the start and the end of a tile fewer
        const int loads_behind = (nsteps - 1 - s < D - 1 ? nsteps - 1 - s : D - 1);
        const int stores_behind = s < D ? s : D;
        const int behind = 4 * (loads_behind + stores_behind);
        // (an immediate: one switch over the few values that occur)
        switch (behind) {
#define MI_CASE(n) case n: wait_vm<n>(); break;
            MI_CASE(0) MI_CASE(4) MI_CASE(8) MI_CASE(12) MI_CASE(16) MI_CASE(20) MI_CASE(24) MI_CASE(28) MI_CASE(32)
#undef MI_CASE
            default: wait_vm<0>(); break;
        }
        fetch(cur, s);
        if (s + D < nsteps) issue(s + D);                     // into the slot just read
        const int64_t r = r0 + (int64_t)s * U;
        const double *cb = col + r;
        if constexpr (MODE != 1 && L8 == 3) {
            if (s == 0) links_first(cb, o1, o2);
#pragma unroll
            for (int i0 = 0; i0 < K; i0 += 8)
                links8c(cur, p[i0], p[i0 + 1], p[i0 + 2], p[i0 + 3], p[i0 + 4], p[i0 + 5], p[i0 + 6], p[i0 + 7],
                        cb + (int64_t)(i0 / 4 + 1) * group_stride,
                        i0 + 8 < K ? cb + (int64_t)(i0 / 4 + 2) * group_stride : cb + U, o1, o2);
        } else if constexpr (MODE != 1 && L8 == 2) {
#pragma unroll
            for (int i0 = 0; i0 < K; i0 += 4)
                links4t8(cur, p[i0], p[i0 + 1], p[i0 + 2], p[i0 + 3], cb + (int64_t)(i0 / 4) * group_stride, o1, o2, o3);
        } else if constexpr (MODE != 1 && L8 == 1) {
#pragma unroll
            for (int i0 = 0; i0 < K; i0 += 8)
                links8(cur, p[i0], p[i0 + 1], p[i0 + 2], p[i0 + 3], p[i0 + 4], p[i0 + 5], p[i0 + 6], p[i0 + 7],
                       cb + (int64_t)(i0 / 4) * group_stride, cb + (int64_t)(i0 / 4 + 1) * group_stride, o1, o2);
        } else if constexpr (MODE != 1) {
#pragma unroll
            for (int i0 = 0; i0 < K; i0 += 4)
                links4(cur, p[i0], p[i0 + 1], p[i0 + 2], p[i0 + 3], cb + (int64_t)(i0 / 4) * group_stride, o1, o2, o3);
        }
        if constexpr (MODE == 2) {
            if (cur[0].x == 1.2345e300) st2(r, cur[0]);       // (keeps the chain alive, never true)
        } else if (r + U <= r1) {
#pragma unroll
            for (int u = 0; u < U; ++u) st2(r + u, cur[u]);
        } else {
            // the matrix's last, partial step: fewer stores than the count above assumes -- nothing is
            // read after it, so nothing depends on the count any more
#pragma unroll
            for (int u = 0; u < U; ++u) if (r + u < r1) st2(r + u, cur[u]);
        }
    }
    if constexpr (L8 == 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

static std::vector<double> g_ref;

template <typename F>
static double time_it(F launch, int reps)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) launch();
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return ms * 1e3 / reps;
}

struct Setup { double *M, *M0, *prow, *col; int64_t rows, ld, cs; int strips, sp; };

static void reset(const Setup &S) { CK(hipMemcpy(S.M, S.M0, S.rows * S.ld * 8, hipMemcpyDeviceToDevice)); }

static void check(const Setup &S, const char *what)
{
    std::vector<double> out((size_t)(S.rows * S.ld));
    CK(hipMemcpy(out.data(), S.M, out.size() * 8, hipMemcpyDeviceToHost));
    if (g_ref.empty()) { g_ref = out; printf("    (%s: reference result kept)\n", what); return; }
    const bool same = memcmp(out.data(), g_ref.data(), out.size() * 8) == 0;
    printf("    %s: %s\n", what, same ? "bit-identical to the reference" : "*** DIFFERS from the reference ***");
}

template <int K, bool NT>
static void run_reg(const Setup &S, int tr)
{
    const dim3 grid((unsigned)S.strips, (unsigned)((S.rows + tr - 1) / tr));
    auto launch = [&]() { hipLaunchKernelGGL((k_reg<K, NT>), grid, dim3(256), 0, 0, S.M, S.prow, S.col, S.ld, S.rows, S.cs, tr, S.sp); };
    reset(S); launch(); CK(hipDeviceSynchronize()); check(S, "registers");
    const double us = time_it(launch, 20);
    hipFuncAttributes fa;
    CK(hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(&k_reg<K, NT>)));
    printf("R  registers, one step ahead      K=%d nt=%d tr=%3d : %8.1f us  %5.2f TB/s   (%d VGPRs, %d B scratch)\n", K, (int)NT, tr, us,
           2.0 * S.rows * S.ld * 8 / us * 1e-6, fa.numRegs, (int)fa.localSizeBytes);
    fflush(stdout);
}

template <int K, int D, bool NT, int MODE, int WMAX, int L8 = 0>
static void run_lds_mode(const Setup &S, int tr, const char *what)
{
    const dim3 grid((unsigned)S.strips, (unsigned)((S.rows + tr - 1) / tr));
    const size_t lds = (size_t)4 * D * 4 * 1024;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_lds<K, D, NT, MODE, WMAX, L8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    auto launch = [&]() { hipLaunchKernelGGL((k_lds<K, D, NT, MODE, WMAX, L8>), grid, dim3(256), lds, 0, S.M, S.prow, S.col, S.ld, S.rows, S.cs, tr, S.sp); };
    if (MODE == 0) { reset(S); launch(); CK(hipDeviceSynchronize()); check(S, what); }
    const double us = time_it(launch, 20);
    hipFuncAttributes fa;
    CK(hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(&k_lds<K, D, NT, MODE, WMAX, L8>)));
    printf("   %-46s D=%d K=%d nt=%d tr=%3d max %d waves/SIMD: %8.1f us  %5.2f TB/s   (%d VGPRs)\n", what, D, K, (int)NT, tr, WMAX, us,
           2.0 * S.rows * S.ld * 8 / us * 1e-6, fa.numRegs);
    fflush(stdout);
}

template <int K, int D, bool NT>
static void run_lds(const Setup &S, int tr)
{
    const dim3 grid((unsigned)S.strips, (unsigned)((S.rows + tr - 1) / tr));
    const size_t lds = (size_t)4 * D * 4 * 1024;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_lds<K, D, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    auto launch = [&]() { hipLaunchKernelGGL((k_lds<K, D, NT>), grid, dim3(256), lds, 0, S.M, S.prow, S.col, S.ld, S.rows, S.cs, tr, S.sp); };
    reset(S); launch(); CK(hipDeviceSynchronize());
    char what[64]; snprintf(what, sizeof what, "LDS ring D=%d tr=%d", D, tr);
    check(S, what);
    const double us = time_it(launch, 20);
    hipFuncAttributes fa;
    CK(hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(&k_lds<K, D, NT>)));
    printf("L  LDS ring, %d steps ahead         K=%d nt=%d tr=%3d : %8.1f us  %5.2f TB/s   (%d VGPRs, %d B scratch, %zu B LDS)\n", D, K, (int)NT, tr, us,
           2.0 * S.rows * S.ld * 8 / us * 1e-6, fa.numRegs, (int)fa.localSizeBytes, lds);
    fflush(stdout);
}

int main(int argc, char **argv)
{
    Setup S;
    S.rows = argc > 1 ? atoll(argv[1]) : 4097; S.ld = argc > 2 ? atoll(argv[2]) : 8208;
    S.cs = (S.rows + 64 + 15) / 16 * 16;
    const int64_t ldv = S.ld / 2;
    S.strips = (int)((ldv + 255) / 256);
    int64_t sp = (ldv + S.strips - 1) / S.strips; sp = (sp + 7) / 8 * 8; if (sp > 256) sp = 256;
    S.sp = (int)sp; S.strips = (int)((ldv + sp - 1) / sp);
    CK(hipMalloc(&S.M, S.rows * S.ld * 8)); CK(hipMalloc(&S.M0, S.rows * S.ld * 8));
    CK(hipMalloc(&S.prow, 32 * S.ld * 8)); CK(hipMalloc(&S.col, 32 * S.cs * 8));
    {   // small random operands: no overflow over 24 links, every bit exercised
        std::vector<double> h((size_t)(S.rows * S.ld));
        uint64_t x = 0x9E3779B97F4A7C15ull;
        auto rnd = [&]() { x += 0x9E3779B97F4A7C15ull; uint64_t z = x; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31; return (double)(z >> 11) * (1.0 / 9007199254740992.0); };
        for (auto &v : h) v = rnd() - 0.5;
        CK(hipMemcpy(S.M0, h.data(), h.size() * 8, hipMemcpyHostToDevice));
        std::vector<double> pr((size_t)(32 * S.ld)), cl((size_t)(32 * S.cs));
        for (auto &v : pr) v = rnd() - 0.5;
        for (auto &v : cl) v = rnd() - 0.5;
        CK(hipMemcpy(S.prow, pr.data(), pr.size() * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(S.col, cl.data(), cl.size() * 8, hipMemcpyHostToDevice));
    }
    printf("tableau %lld x %lld doubles = %.3f GB stored, %d strips of %d pairs\n", (long long)S.rows, (long long)S.ld, S.rows * S.ld * 8 / 1e9, S.strips, S.sp);
    const bool big = S.rows * S.ld * 8 > (320ll << 20);
#define BOTH(call_nt, call_plain) do { if (big) { call_nt; } else { call_plain; } } while (0)
    for (int tr : {32, 64}) {
        BOTH((run_reg<24, true>(S, tr)), (run_reg<24, false>(S, tr)));
        BOTH((run_lds<24, 2, true>(S, tr)), (run_lds<24, 2, false>(S, tr)));
        BOTH((run_lds<24, 3, true>(S, tr)), (run_lds<24, 3, false>(S, tr)));
    }
    // where does the time go?  the same kernel without the links (memory only) and without the global
    // loads and stores (links + their scalar operand loads only), at the product's occupancy and free
    BOTH((run_lds_mode<24, 2, true, 1, 3>(S, 32, "memory only (no links), 3 waves per SIMD")), (run_lds_mode<24, 2, false, 1, 3>(S, 32, "memory only (no links), 3 waves per SIMD")));
    BOTH((run_lds_mode<24, 2, true, 1, 8>(S, 32, "memory only (no links), occupancy free")), (run_lds_mode<24, 2, false, 1, 8>(S, 32, "memory only (no links), occupancy free")));
    BOTH((run_lds_mode<24, 2, true, 2, 3>(S, 32, "links only (no global traffic), 3 waves per SIMD")), (run_lds_mode<24, 2, false, 2, 3>(S, 32, "links only (no global traffic), 3 waves per SIMD")));
    BOTH((run_lds_mode<24, 2, true, 2, 8>(S, 32, "links only (no global traffic), occupancy free")), (run_lds_mode<24, 2, false, 2, 8>(S, 32, "links only (no global traffic), occupancy free")));
    BOTH((run_lds_mode<24, 2, true, 0, 3>(S, 32, "everything, 3 waves per SIMD")), (run_lds_mode<24, 2, false, 0, 3>(S, 32, "everything, 3 waves per SIMD")));
    BOTH((run_lds_mode<24, 2, true, 0, 4>(S, 32, "everything, up to 4 waves per SIMD")), (run_lds_mode<24, 2, false, 0, 4>(S, 32, "everything, up to 4 waves per SIMD")));
    BOTH((run_lds_mode<24, 2, true, 2, 3, 1>(S, 32, "8 links per statement: links only, 3 waves")), (run_lds_mode<24, 2, false, 2, 3, 1>(S, 32, "8 links per statement: links only, 3 waves")));
    BOTH((run_lds_mode<24, 2, true, 0, 3, 1>(S, 32, "8 links per statement: everything, 3 waves")), (run_lds_mode<24, 2, false, 0, 3, 1>(S, 32, "8 links per statement: everything, 3 waves")));
    BOTH((run_lds_mode<24, 2, true, 0, 4, 1>(S, 32, "8 links per statement: everything, <= 4 waves")), (run_lds_mode<24, 2, false, 0, 4, 1>(S, 32, "8 links per statement: everything, <= 4 waves")));
    BOTH((run_lds_mode<24, 3, true, 0, 3, 1>(S, 32, "8 links per statement: everything, D=3")), (run_lds_mode<24, 3, false, 0, 3, 1>(S, 32, "8 links per statement: everything, D=3")));
    BOTH((run_lds_mode<24, 2, true, 2, 3, 2>(S, 32, "8 temporaries: links only, 3 waves")), (run_lds_mode<24, 2, false, 2, 3, 2>(S, 32, "8 temporaries: links only, 3 waves")));
    BOTH((run_lds_mode<24, 2, true, 0, 3, 2>(S, 32, "8 temporaries: everything, 3 waves")), (run_lds_mode<24, 2, false, 0, 3, 2>(S, 32, "8 temporaries: everything, 3 waves")));
    BOTH((run_lds_mode<24, 2, true, 2, 3, 3>(S, 32, "continuous chain: links only, 3 waves")), (run_lds_mode<24, 2, false, 2, 3, 3>(S, 32, "continuous chain: links only, 3 waves")));
    BOTH((run_lds_mode<24, 2, true, 0, 3, 3>(S, 32, "continuous chain: everything, 3 waves")), (run_lds_mode<24, 2, false, 0, 3, 3>(S, 32, "continuous chain: everything, 3 waves")));
    BOTH((run_lds_mode<24, 2, true, 0, 4, 3>(S, 32, "continuous chain: everything, <= 4 waves")), (run_lds_mode<24, 2, false, 0, 4, 3>(S, 32, "continuous chain: everything, <= 4 waves")));
    BOTH((run_lds_mode<24, 2, true, 0, 3, 1>(S, 32, "8 links per statement again: everything, 3 waves")), (run_lds_mode<24, 2, false, 0, 3, 1>(S, 32, "8 links per statement again: everything, 3 waves")));
    // the other traffic policy at this size, for the record
    BOTH((run_lds<24, 3, false>(S, 64)), (run_lds<24, 3, true>(S, 64)));
    return 0;
}
