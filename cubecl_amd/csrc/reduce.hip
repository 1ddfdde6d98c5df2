// reduce.hip -- array-wide and last-axis reductions for gfx950 (HBM-bound streaming kernels).
//
// Roofline: HBM read bandwidth (8.0 TB/s spec).  Algorithmic traffic = 4 bytes per element read
// once (+16 B per workgroup of partials, ignored).  Design rules followed
// (/opt/skills/guides/cdna_hip_programming.md G2/G7/G11/G13, Appendix B "Reduction"):
//   - 16 B per lane coalesced loads (global_load_dwordx4, non-temporal: the data is read once),
//     RED_UNROLL independent loads in flight per lane, one independent accumulator per load slot;
//   - grid = ONE workgroup per CU since round 5 (the loop below keeps 8-16 loads of every lane in flight by hand; with
//     the compiler's schedule of rounds 1-4 -- one load, a wait, seven loads -- it took 3 per CU).  128 MiB / 1 GiB, us,
//     at 1 / 2 / 3 per CU: sum 23.2 / 23.7 / 24.5 and 154.8 / 154.6 / 157.0; fused sum + argmax 26.3 / 26.7 / 27.6 and
//     155.0 / 162.1 / 166.1 (profiles/r05_c4_shard.md); fewer workgroups = fewer partial records and arrival atomics at
//     the tail.  Grid-stride over 32 KiB tiles (no XCD remap: there is no inter-workgroup reuse, guide T1 "Transfer: 0% on
//     LayerNorm"), the remainder dealt out in 4 KiB rows so that every workgroup streams the same bytes +- one row;
//   - wave64 xor butterfly in the reference's plane_reduce order
//     (crates/cubecl-cpp/src/shared/plane.rs:60-70), then LDS across the 4 waves, one partial
//     record per workgroup; the LAST workgroup to arrive (ticket word) folds the records in index
//     order inside the same launch: the summation tree is a pure function of (n, grid) =>
//     bit-reproducible run to run, no float atomics, and no second launch (the separate fold
//     kernel + stream boundary cost ~6 us of a ~160 us pass);
//   - inter-workgroup hand-off by 8-byte agent-scope atomics on both sides (write-through sc1
//     stores, per-wave vmcnt(0) drain, then the ticket; sc1 loads in the folding workgroup) --
//     placement independent (guide section 6, Guideline 16).  The ticket word is library-owned
//     scratch (one per stream), zero between calls because the last arriver resets it.
#include "internal.hpp"

#include <algorithm>
#include <type_traits>
#include <cstdlib>

using namespace mi355;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4r __attribute__((ext_vector_type(4)));

namespace {

constexpr int RED_BLOCK = 256;                              // 4 waves
constexpr int RED_UNROLL = 8;                               // 16-B loads in flight per lane
constexpr int RED_TILE = RED_BLOCK * RED_UNROLL * 4;        // 8192 floats = 32 KiB per tile
constexpr int RED_MAX_GRID = 4096;

// Input element types of the array-wide reductions: f32, or bf16 / f16 widened to f32 on load (exact), always 16 bytes
// per lane and load, sums and compares in f32.  EPV = elements per 16-byte vector.
template <int DT> struct red_in;
template <> struct red_in<MI355_DTYPE_F32> {
    typedef float elem;
    static constexpr int EPV = 4;
    static __device__ __forceinline__ float widen(float v) { return v; }
    static __device__ __forceinline__ void unpack(const u32x4r &raw, float (&v)[4])
    {
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = __uint_as_float(raw[c]);
    }
};
template <> struct red_in<MI355_DTYPE_BF16> {
    typedef uint16_t elem;
    static constexpr int EPV = 8;
    static __device__ __forceinline__ float widen(uint16_t v) { return __uint_as_float((uint32_t)v << 16); }
    static __device__ __forceinline__ void unpack(const u32x4r &raw, float (&v)[8])
    {
#pragma unroll
        for (int c = 0; c < 4; ++c) { v[2 * c] = __uint_as_float(raw[c] << 16); v[2 * c + 1] = __uint_as_float(raw[c] & 0xFFFF0000u); }
    }
};
template <> struct red_in<MI355_DTYPE_F16> {
    typedef uint16_t elem;
    static constexpr int EPV = 8;
    static __device__ __forceinline__ float widen(uint16_t v)
    {
        _Float16 h;
        __builtin_memcpy(&h, &v, 2);
        return (float)h;
    }
    static __device__ __forceinline__ void unpack(const u32x4r &raw, float (&v)[8])
    {
#pragma unroll
        for (int c = 0; c < 4; ++c) { v[2 * c] = widen((uint16_t)(raw[c] & 0xFFFFu)); v[2 * c + 1] = widen((uint16_t)(raw[c] >> 16)); }
    }
};

// Dev timing trace (-DRED_TRACE, never in the product library): 100 MHz-counter stamps of thread 0 of every workgroup -- entry,
// whole rounds done, dealt rows done, record stored, ticket taken, exit -- read back with mi355_dev_red_trace
// (tools/dev/reduce_trace.py).
#ifdef RED_TRACE
__device__ unsigned long long red_trace_buf[4096 * 8];
#define RED_STAMP(slot) do { if (threadIdx.x == 0) red_trace_buf[blockIdx.x * 8 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define RED_STAMP(slot) do { } while (0)
#endif

struct __attribute__((aligned(16))) red_record {
    float sum;
    uint32_t key;
    uint64_t idx;
};

// Order-preserving key for the argmax rule (oracle/oracle.c argmax_key): IEEE order, -0 == +0,
// NaN above everything.
__device__ __forceinline__ uint32_t argmax_key(float v)
{
    uint32_t u = __float_as_uint(v);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return 0xFFFFFFFFu;
    if (u == 0x80000000u) u = 0u;
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// (key, idx) combine: larger key wins, equal keys keep the LOWER index.
__device__ __forceinline__ void arg_combine(uint32_t &key, uint64_t &idx, uint32_t okey, uint64_t oidx)
{
    const bool take = (okey > key) || (okey == key && oidx < idx);
    key = take ? okey : key;
    idx = take ? oidx : idx;
}

__device__ __forceinline__ void arg_combine_u32(uint32_t &key, uint32_t &idx, uint32_t okey, uint32_t oidx)
{
    const bool take = (okey > key) || (okey == key && oidx < idx);
    key = take ? okey : key;
    idx = take ? oidx : idx;
}

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// ---- the reduce operations (round 4: max / min / prod / mean values and argmin beside sum and argmax) -----------------------
// Value operations VOP (the public MI355_REDUCE_* codes; MEAN is SUM with the final division) and index operations AOP.
//   MAX / MIN  the maximum / minimum under IEEE comparison with -0 < +0; NaN if ANY element is NaN (numpy's max / min; the
//              value at the argmax / argmin below, up to the NaN's payload and the sign of a zero tie).  The NaN test travels
//              beside the running value as a flag (v_max_f32 / v_min_f32 drop NaNs), so the result is order independent.
//   PROD       f32 product in the summation's tree shape (plane_prod, crates/cubecl-core/src/frontend/plane.rs:285)
//   ARGMIN     the mirror image of ARGMAX: lowest index of the minimum, -0 == +0, NaN ranks FIRST and the first NaN wins
//              (numpy's argmin) -- the argmax key mirrored (~key) with the NaN code kept on top, same combine
enum { VOP_NONE = -1, VOP_SUM = MI355_REDUCE_SUM, VOP_MAX = MI355_REDUCE_MAX, VOP_MIN = MI355_REDUCE_MIN, VOP_PROD = MI355_REDUCE_PROD };
enum { AOP_NONE = 0, AOP_MAX = 1, AOP_MIN = 2 };

template <int VOP> struct vop {
    static __device__ __forceinline__ float identity()
    { return VOP == VOP_PROD ? 1.f : VOP == VOP_MAX ? -__builtin_inff() : VOP == VOP_MIN ? __builtin_inff() : 0.f; }
    static __device__ __forceinline__ float apply(float a, float b)
    {
        if constexpr (VOP == VOP_PROD) return a * b;
        else if constexpr (VOP == VOP_MAX) return __builtin_fmaxf(a, b);
        else if constexpr (VOP == VOP_MIN) return __builtin_fminf(a, b);
        else return a + b;
    }
    static __device__ __forceinline__ f32x4 apply(f32x4 a, f32x4 b)
    {
        if constexpr (VOP == VOP_PROD) return a * b;
        else if constexpr (VOP == VOP_MAX) return __builtin_elementwise_max(a, b);
        else if constexpr (VOP == VOP_MIN) return __builtin_elementwise_min(a, b);
        else return a + b;
    }
    static constexpr bool TRACKS_NAN = VOP == VOP_MAX || VOP == VOP_MIN;
};

// One step of the wave64 xor butterfly WITHOUT the LDS crossbar (round 6).  __shfl_xor is ds_bpermute_b32: ~130 cycles each, and the
// folds below are chains of six of them -- two folds of the sum and two of {key, index, bits} stood for ~1.5 us of the 2.7 us between
// the last byte of a 128 MiB fused pass and its result.  Step s pairs lane i with lane i ^ 2^s; every operation folded here is
// symmetric (a + b, a * b, max, min bit for bit -- v_max / v_min order -0 below +0 -- and the (key, index) rule), so after step s all
// lanes of a group of 2^(s+1) hold the same value and ANY lane of the partner group may stand for lane i ^ 2^s:
//   steps 0, 1: quad-permute DPP (the exact partner); steps 2, 3: row_half_mirror / row_mirror DPP (a lane of the partner group);
//   steps 4, 5: v_permlane16_swap / v_permlane32_swap of the value with itself: a = {rows 0,0,2,2} / {lower half twice},
//   b = {rows 1,1,3,3} / {upper half twice} -- own and partner in one order or the other.
// Returns (a, b) = the two values to combine: the same pairs as the xor butterfly of the reference's plane_reduce lowering
// (crates/cubecl-cpp/src/shared/plane.rs:60-70), hence the same bits as rounds 1-5.
template <int STEP>
__device__ __forceinline__ void butterfly_pair(uint32_t v, uint32_t &a, uint32_t &b)
{
    if constexpr (STEP == 0) { a = v; b = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true); }          // quad_perm [1,0,3,2]
    else if constexpr (STEP == 1) { a = v; b = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true); }     // quad_perm [2,3,0,1]
    else if constexpr (STEP == 2) { a = v; b = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x141, 0xF, 0xF, true); }    // row_half_mirror
    else if constexpr (STEP == 3) { a = v; b = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x140, 0xF, 0xF, true); }    // row_mirror
    else if constexpr (STEP == 4) { const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false); a = r[0]; b = r[1]; }
    else { const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false); a = r[0]; b = r[1]; }
}

template <int VOP, int STEP>
__device__ __forceinline__ float wave_fold_step(float v)
{
    uint32_t a, b;
    butterfly_pair<STEP>(__float_as_uint(v), a, b);
    return vop<VOP>::apply(__uint_as_float(a), __uint_as_float(b));
}

template <int VOP>
__device__ __forceinline__ float wave_fold(float v)
{
    v = wave_fold_step<VOP, 0>(v); v = wave_fold_step<VOP, 1>(v); v = wave_fold_step<VOP, 2>(v);
    v = wave_fold_step<VOP, 3>(v); v = wave_fold_step<VOP, 4>(v); v = wave_fold_step<VOP, 5>(v);
    return v;
}

// key of the index operations: larger key wins in arg_combine; ARGMIN mirrors every number's key and keeps NaN on top
template <int AOP>
__device__ __forceinline__ uint32_t arg_key(float v)
{
    const uint32_t k = argmax_key(v);
    return (AOP == AOP_MIN && k != 0xFFFFFFFFu) ? ~k : k;
}

// (key, idx) combine carrying the winner's value bits along (the array-wide kernel reports the winning element bit-exactly
// without going back to memory for it: key collapses -0 / +0 and NaN payloads)
__device__ __forceinline__ void arg_combine3(uint32_t &key, uint64_t &idx, uint32_t &bits, uint32_t okey, uint64_t oidx, uint32_t obits)
{
    const bool take = (okey > key) || (okey == key && oidx < idx);
    key = take ? okey : key;
    idx = take ? oidx : idx;
    bits = take ? obits : bits;
}

template <int STEP>
__device__ __forceinline__ void wave_argmax3_step(uint32_t &key, uint64_t &idx, uint32_t &bits)
{
    uint32_t ka, kb, ba, bb, la, lb, ha, hb;
    butterfly_pair<STEP>(key, ka, kb);
    butterfly_pair<STEP>(bits, ba, bb);
    butterfly_pair<STEP>((uint32_t)idx, la, lb);
    butterfly_pair<STEP>((uint32_t)(idx >> 32), ha, hb);
    key = ka; bits = ba; idx = ((uint64_t)ha << 32) | la;
    arg_combine3(key, idx, bits, kb, ((uint64_t)hb << 32) | lb, bb);        // (symmetric: larger key, then lower index; two lanes never tie on both)
}

__device__ __forceinline__ void wave_argmax3(uint32_t &key, uint64_t &idx, uint32_t &bits)
{
    wave_argmax3_step<0>(key, idx, bits); wave_argmax3_step<1>(key, idx, bits); wave_argmax3_step<2>(key, idx, bits);
    wave_argmax3_step<3>(key, idx, bits); wave_argmax3_step<4>(key, idx, bits); wave_argmax3_step<5>(key, idx, bits);
}

__device__ __forceinline__ void wave_argmax(uint32_t &key, uint64_t &idx)
{
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t okey = __shfl_xor(key, off, 64);
        const uint32_t olo = __shfl_xor((uint32_t)idx, off, 64);
        const uint32_t ohi = __shfl_xor((uint32_t)(idx >> 32), off, 64);
        arg_combine(key, idx, okey, ((uint64_t)ohi << 32) | olo);
    }
}

typedef __attribute__((address_space(1))) unsigned long long gu64;

__device__ __forceinline__ void record_store(red_record *slot, float sum, uint32_t key, uint64_t idx)
{
    gu64 *p = (gu64 *)(unsigned long long *)slot;
    __hip_atomic_store(p, ((unsigned long long)key << 32) | __float_as_uint(sum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(p + 1, idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ red_record record_load(const red_record *slot)
{
    gu64 *p = (gu64 *)(unsigned long long *)const_cast<red_record *>(slot);
    const unsigned long long a = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long b = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    red_record r;
    r.sum = __uint_as_float((uint32_t)a);
    r.key = (uint32_t)(a >> 32);
    r.idx = b;
    return r;
}

// Arrival tickets: returning agent-scope fetch_adds (a compare-and-swap loop here costs one L2 round trip per CONTENDER:
// 2048 workgroups finishing together took 8 ms).  The words live in library-owned device scratch that is zero between
// calls: whoever completes a count puts the zero back.
//   Two levels since round 5.  One L2 serves the fetch_adds on ONE address at 11.4 ns each, whatever the number of waves
//   asking (tools/dev/atomic_rate_probe.hip: 8 addresses 256 bytes apart 1.8 ns, 64 addresses 0.3 ns): with one word, the 768
//   workgroups of a pass that finish together -- the 128 MiB shard of config C4 streams for 19 us and its workgroups finish
//   within 3 us of each other -- queued for 768 x 11.4 ns = 8.8 us behind that word, a third of the pass (trace in
//   profiles/r05_c4_shard.md).  Now workgroup b counts on word b % 32 of 32 group words 256 bytes apart, and the last of
//   a group counts on the top word: at most 24 + 32 serialised adds on the way of any workgroup.
constexpr uint32_t RED_TICKET_GROUPS = 32;
constexpr uint32_t RED_GROUP_WORD0 = 512, RED_GROUP_WORD_STEP = 64;     // words: the groups start 2 KiB into the stream's slot
__device__ __forceinline__ bool arrive_is_last(unsigned int *ticket, uint32_t G)
{
    typedef __attribute__((address_space(1))) unsigned int gu32;
    const uint32_t g = blockIdx.x % RED_TICKET_GROUPS;
    const uint32_t members = (G - g + RED_TICKET_GROUPS - 1) / RED_TICKET_GROUPS;
    gu32 *grp = (gu32 *)(ticket + RED_GROUP_WORD0 + g * RED_GROUP_WORD_STEP);
    if (__hip_atomic_fetch_add(grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != members - 1) return false;
    __hip_atomic_store(grp, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);                   // ready for the next call
    const uint32_t groups = G < RED_TICKET_GROUPS ? G : RED_TICKET_GROUPS;
    return __hip_atomic_fetch_add((gu32 *)ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == groups - 1;
}

// Round 6: the hand-off WITHOUT tickets for grids of at most RED_POLL_MAX workgroups (the default grid: one per CU).  A ticket costs the
// tail of a 128 MiB pass three dependent L2 round trips after a workgroup's last byte -- the record store must be acknowledged before
// the ticket is taken (vmcnt(0)), the group ticket, the top ticket -- and then the folding workgroup loads the records: 2.6 us behind
// an 18.3 us stream (profiles/r05_c4_shard.md).  Instead every workgroup stores its record as ONE 16-byte write-through store whose
// last bit says "valid" and leaves; workgroup G - 1 -- dispatched last, so every other workgroup is running or done: it can wait
// without holding anybody up -- polls the G records (thread t owns t, t + 256, ...: one 16-byte agent-scope load per poll), takes each
// the moment its valid bit shows, puts the bit back to zero for the next call, and folds in index order as before: same tree, same
// bits.  One store + one load round trip behind the last byte.  The records live in library-owned device scratch (the stream's
// ticket slot, zero between calls like the tickets: a caller's workspace could hold anything).  A 16-byte aligned dwordx4 store /
// load is one L2 transaction each, so a record is seen whole or not at all (stress: tests/test_gpu_reduce.py).
constexpr uint32_t RED_POLL_WORD0 = 2560, RED_POLL_MAX = 384;           // words: bytes 10240 .. 16383 of the stream's 16 KiB slot
constexpr uint64_t RED_VALID = 1ull << 63;                              // (indices stay below 2^63; an empty shard's ~0 carries the bit anyway)
__device__ __forceinline__ void poll_record_store(red_record *slot, float sum, uint32_t key, uint64_t idx)
{
    const uint64_t tagged = idx | RED_VALID;
    const u32x4r v = {__float_as_uint(sum), key, (uint32_t)tagged, (uint32_t)(tagged >> 32)};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(slot), "v"(v) : "memory");
}
__device__ __forceinline__ red_record poll_record_take(red_record *slot)
{
    u32x4r v;
    for (;;) {
        asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(slot) : "memory");
        if (v[3] & 0x80000000u) break;
        __builtin_amdgcn_s_sleep(1);
    }
    typedef __attribute__((address_space(1))) unsigned long long gu64p;
    __hip_atomic_store((gu64p *)((unsigned long long *)slot + 1), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // invalid again: ready for the next call
    red_record r;
    r.sum = __uint_as_float(v[0]);
    r.key = v[1];
    const uint64_t raw = ((uint64_t)v[3] << 32) | v[2];
    r.idx = raw == ~0ull ? ~0ull : (raw & ~RED_VALID);
    return r;
}

// Every workgroup folds its tiles into one record; the last one to arrive (tickets) / workgroup G - 1 (polling) folds the G records.
//   in      : 16-byte aligned body of the array (host peels a misaligned head into `head`)
//   head    : up to 3 leading elements (global indices 0..head_n-1), body index i maps to
//             global index i + head_n
//   VOP / AOP : value and index operation (above); both at once only as SUM + ARGMAX, the fused pass of the sharded job.
//   mean_div  : MEAN = SUM with out = sum / mean_div (0 = no division)
template <int VOP, int AOP, int DT = MI355_DTYPE_F32>
__global__ void __launch_bounds__(RED_BLOCK) __attribute__((amdgpu_waves_per_eu(1, 8)))   // one workgroup per CU is the grid (pick_grid): the whole register file is one wave's
reduce_kernel(const typename red_in<DT>::elem *__restrict__ head, uint32_t head_n, const typename red_in<DT>::elem *__restrict__ in, uint64_t n,
              red_record *__restrict__ records, unsigned int *__restrict__ ticket, uint64_t n_total, float *__restrict__ out_sum, float *__restrict__ out_val, uint64_t *__restrict__ out_idx,
              float mean_div, uint64_t rounds, uint32_t poll)
{
    constexpr bool SUM = VOP != VOP_NONE, ARG = AOP != AOP_NONE;        // (SUM: "a value is folded", whatever the operation)
    static_assert(!(SUM && ARG) || (VOP == VOP_SUM && AOP == AOP_MAX), "fused pass: sum + argmax only");
    typedef vop<VOP == VOP_NONE ? VOP_SUM : VOP> V;
    constexpr bool NANS = SUM && V::TRACKS_NAN;
    constexpr bool AMIN = AOP == AOP_MIN;
    typedef red_in<DT> RI;
    constexpr int EPV = RI::EPV;
    constexpr uint64_t ROW = (uint64_t)RED_BLOCK * EPV;                     // elements per 4 KiB row (one 16-byte load per lane)
    const uint32_t tid = threadIdx.x;
    const uint32_t G = gridDim.x;

    // The first tile's loads are the first thing the kernel does (round 6): everything below -- accumulator set-up, the head, the deal of
    // the remainder rows (two 32-bit divisions) -- runs while they are on their way instead of in front of them (first bytes of a cold
    // 128 MiB pass: 2 us after the first workgroup's entry, of which ~0.8 us is the launch ramp of 256 workgroups).
    u32x4r ra[RED_UNROLL], rb[RED_UNROLL];
    if (rounds) {
        const u32x4r *__restrict__ v0 = reinterpret_cast<const u32x4r *>(in) + (uint64_t)blockIdx.x * (RED_UNROLL * RED_BLOCK) + tid;
#pragma unroll
        for (int u = 0; u < RED_UNROLL; ++u) ra[u] = __builtin_nontemporal_load(v0 + (uint64_t)u * RED_BLOCK);
    }
    __builtin_amdgcn_sched_barrier(0);

    f32x4 acc[RED_UNROLL];
#pragma unroll
    for (int u = 0; u < RED_UNROLL; ++u) acc[u] = (f32x4){V::identity(), V::identity(), V::identity(), V::identity()};
    float tail_acc = V::identity();
    bool nan_seen = false;           // MAX / MIN: some element of this lane's share was a NaN
    uint32_t best_key = 0u;          // 0 = "nothing yet": every real key is >= 0x007FFFFF (-inf; ARGMIN: +inf)
    uint64_t best_idx = ~0ull;       // (head and ragged tail elements; the streamed vectors join them below)
    uint32_t best_bits = 0u;         // the element behind (best_key, best_idx), as loaded (widened)
    constexpr float NAN_LEADS = AMIN ? -__builtin_inff() : __builtin_inff();
    // Index operations over the streamed vectors, branch-free per vector since round 5: the running extremum under FLOAT
    // comparison (-0 == +0, a NaN never enters), the vector that first reached it (kept whole, with its item / slot code),
    // and the position of the lane's first NaN; resolved to (key, index) once, behind the stream.  (Rounds 1-4 rejected a
    // vector with one compare and took an exact per-element path otherwise -- but a wave takes that path when ANY of its 64
    // lanes has a new extremum, which at the 64 vectors a lane sees of a 128 MiB shard is nearly every vector.)
    float s_best = -NAN_LEADS;
    uint32_t s_code = ~0u;           // item * 8 + slot; ~0 = nothing streamed yet
    u32x4r s_raw = (u32x4r){0u, 0u, 0u, 0u};
    uint32_t s_nan = ~0u;            // (item * 8 + slot) * EPV + position of the first NaN; ~0 = none
    uint32_t s_nan_bits = 0u;        // that NaN

    // peeled head: lowest global indices, block 0 only
    if (blockIdx.x == 0 && tid < head_n) {
        const float v = RI::widen(head[tid]);
        if (SUM) tail_acc = V::apply(tail_acc, v);
        if (NANS) nan_seen |= (v != v);
        if (ARG) { best_key = arg_key<AOP>(v); best_idx = tid; best_bits = __float_as_uint(v); }
    }

    // One 16-byte vector of this lane: `code` = item * 8 + slot names it (item_base(item) + slot * 256 = its index among the
    // body's vectors; global element = that * EPV + head_n).
    // `live` (wave-uniform; constant true in the steady loop): a spare load of a short last item contributes nothing -- the
    // registers are READ all the same (a load result that is dead on some path makes the compiler's waitcnt pass drain
    // every load in flight before the register is written again).
    auto consume = [&](auto guarded, const bool live, const u32x4r &rawv, f32x4 &accu, const uint32_t code) __attribute__((always_inline)) {
        constexpr bool GUARD = decltype(guarded)::value;
        float w[EPV];
        RI::unpack(rawv, w);
        if (SUM) {
            if constexpr (GUARD) {
#pragma unroll
                for (int c = 0; c < EPV; ++c) w[c] = live ? w[c] : V::identity();
            }
            accu = V::apply(accu, (f32x4){w[0], w[1], w[2], w[3]});
            if constexpr (EPV == 8) accu = V::apply(accu, (f32x4){w[4], w[5], w[6], w[7]});
        }
        if constexpr (NANS) {               // (the identity is never a NaN)
            nan_seen |= __builtin_isunordered(w[0], w[1]) | __builtin_isunordered(w[2], w[3]);
            if constexpr (EPV == 8) nan_seen |= __builtin_isunordered(w[4], w[5]) | __builtin_isunordered(w[6], w[7]);
        }
        if (ARG) {
            typedef vop<AMIN ? VOP_MIN : VOP_MAX> X;
            float m4 = X::apply(X::apply(w[0], w[1]), X::apply(w[2], w[3]));                 // (v_max / v_min drop NaNs)
            bool has_nan = __builtin_isunordered(w[0], w[1]) | __builtin_isunordered(w[2], w[3]);
            if constexpr (EPV == 8) {
                m4 = X::apply(m4, X::apply(X::apply(w[4], w[5]), X::apply(w[6], w[7])));
                has_nan |= __builtin_isunordered(w[4], w[5]) | __builtin_isunordered(w[6], w[7]);
            }
            // strict comparison: within one lane codes only grow, so the first vector to reach the extremum is kept
            // (SUM + ARG guarded: w[] holds the sum's identity for a spare load, and `live` keeps it out of the race)
            const bool take = ((AMIN ? (m4 < s_best) : (m4 > s_best)) | (s_code == ~0u)) & (!GUARD || live);
            s_best = take ? X::apply(s_best, m4) : s_best;
            s_code = take ? code : s_code;
#pragma unroll
            for (int c = 0; c < 4; ++c) s_raw[c] = take ? rawv[c] : s_raw[c];
            if (has_nan & (!GUARD || live)) {                    // rare: exec-masked, skipped by a wave without a NaN
                if (s_nan == ~0u) {
                    uint32_t pos = 0;
#pragma unroll
                    for (int c = EPV - 1; c >= 0; --c)
                        if (w[c] != w[c]) { pos = (uint32_t)c; s_nan_bits = __float_as_uint(w[c]); }
                    s_nan = code * EPV + pos;
                }
            }
        }
    };

    // The body in 4 KiB rows (256 lanes x 16 bytes), cut into ITEMS of a workgroup: whole rounds of 32 KiB tiles (8 rows)
    // grid-stride over the array, and the rows that do not fill another round for EVERY workgroup are dealt out evenly as
    // one last item, a contiguous run of at most 8 rows (round 5: until then the last round was whole tiles for the first
    // `tiles % G` workgroups -- at the 128 MiB shard of config C4, 4096 tiles on 768 workgroups, a third of the workgroups
    // streamed a sixth tile while the rest of the chip idled).
    //
    // Items are double-buffered BY HAND: item i + 1 is on its way while item i is consumed, 8 to 16 loads in flight per
    // lane at any time.  (Left to itself the compiler issued one load, waited for it, then issued the other seven: two
    // exposed memory latencies per tile -- the ISA of rounds 1-4.)  The rules of gemm_nnrows.hip's ring apply: every load
    // unconditional (a short item re-reads its last row: same cache lines, no HBM traffic), no dead load result, constant
    // register indices, the order pinned by sched_barrier; the compiler's own waitcnt pass then counts vmcnt(8).
    const u32x4r *__restrict__ vin = reinterpret_cast<const u32x4r *>(in);
    const uint64_t rows = n / ROW;
    // (`rounds` = rows / (8 G) comes from the host, and the deal below is 32-bit: a 64-bit division is ~150 instructions here)
    const uint64_t row0 = rounds * G * RED_UNROLL;
    const uint32_t rem = (uint32_t)(rows - row0);                                      // rem < 8 G <= 2^15
    const uint32_t lo = rem * blockIdx.x / G, hi = rem * (blockIdx.x + 1u) / G;        // at most 8 rows, wave-uniform
    const uint32_t dealt = hi - lo;
    const uint64_t items = rounds + (dealt ? 1u : 0u);
    // item i: first vector of this lane and live rows (0 = no such item: its loads re-read item 0's first row)
    // (Round 6, measured and dropped -- profiles/r06_shard_tile_rotation.txt, r06_shard_weighted_split.txt: the XCDs of a pass finish up to
    //  5 us apart -- the odd XCD of every pair streams ~9 % slower, and the XCDs are started 0.1-1.2 us apart -- but neither rotating the
    //  tile column by round (so that an XCD does not stay on one eighth of the channels) nor contiguous runs weighted 1000 : 915 against
    //  the odd XCDs moved the pass: the XCDs then finish within 4 % of each other and the pass takes the same 22.7-23.2 us -- the early
    //  finishers' bandwidth goes to the late ones, the array leaves HBM at ~7.3 TB/s either way -- and contiguous runs cost 1-3 % at 1 GiB.)
    auto item_base = [&](uint64_t i) -> uint64_t {
        return (i < rounds ? (i * G + blockIdx.x) * (uint64_t)RED_UNROLL : (i < items ? row0 + lo : (rounds ? (uint64_t)blockIdx.x * RED_UNROLL : row0 + lo))) * RED_BLOCK + tid;
    };
    auto item_rows = [&](uint64_t i) -> uint32_t { return i < rounds ? (uint32_t)RED_UNROLL : (i < items ? dealt : 0u); };
    auto issue_whole = [&](u32x4r (&r)[RED_UNROLL], uint64_t vbase) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < RED_UNROLL; ++u) r[u] = __builtin_nontemporal_load(vin + vbase + (uint64_t)u * RED_BLOCK);
    };
    auto issue_any = [&](u32x4r (&r)[RED_UNROLL], uint64_t vbase, uint32_t cnt) __attribute__((always_inline)) {
        const uint32_t top = cnt ? cnt - 1u : 0u;
#pragma unroll
        for (int u = 0; u < RED_UNROLL; ++u) r[u] = __builtin_nontemporal_load(vin + vbase + (uint64_t)((uint32_t)u < top ? (uint32_t)u : top) * RED_BLOCK);
    };
    RED_STAMP(0);
    if (items) {
        uint64_t i = 0;
        if (!rounds) issue_any(ra, item_base(0), item_rows(0));      // (with whole rounds item 0 is a whole tile and already on its way)
        // steady state: items i and i + 1 are whole tiles.  (Not for the fused pass over 16-bit input: two unguarded
        // copies of its eight-element bodies beside the guarded one took 255 VGPRs and spilled; it runs every item through
        // the turn-by-turn loop below, which double-buffers too -- with 32 register copies per item.)
        // (Refilling each slot the moment it has been consumed -- 15-16 loads in flight instead of 8-16 -- measured SLOWER:
        // sum 24.6 against 23.4 us on the 128 MiB shard, fused 26.9 against 26.0; profiles/r05_c4_shard.md.)
        constexpr bool STEADY = !(SUM && ARG && EPV == 8);
        // (Round 6: the loop runs while items i and i + 1 are whole tiles and fetches item i + 2 with issue_any -- a whole tile, the dealt rows
        // or nothing.  Until then it stopped while i + 2 was still whole and left the last two whole tiles -- an eighth of a 128 MiB shard -- to
        // the guarded turn-by-turn loop below: 2.2 us from "whole rounds done" to "dealt rows done" on the fused pass.)
        for (; STEADY && i + 1 < rounds; i += 2) {
            __builtin_amdgcn_sched_barrier(0);
            issue_whole(rb, item_base(i + 1));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < RED_UNROLL; ++u) { consume(std::false_type{}, true, ra[u], acc[u], (uint32_t)i * RED_UNROLL + u); if constexpr (EPV == 8) __builtin_amdgcn_sched_barrier(0); }
            __builtin_amdgcn_sched_barrier(0);
            issue_any(ra, item_base(i + 2), item_rows(i + 2));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < RED_UNROLL; ++u) { consume(std::false_type{}, true, rb[u], acc[u], (uint32_t)(i + 1) * RED_UNROLL + u); if constexpr (EPV == 8) __builtin_amdgcn_sched_barrier(0); }
        }
        RED_STAMP(1);
        // the last one to three items (the dealt rows among them), one per turn: `ra` holds item i, item i + 1 is fetched
        // into `rb` while it is consumed, then takes its place (a register copy: this loop runs at most three times)
#pragma nounroll
        for (; i < items; ++i) {
            const uint32_t ca = item_rows(i);
            __builtin_amdgcn_sched_barrier(0);
            issue_any(rb, item_base(i + 1), item_rows(i + 1));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < RED_UNROLL; ++u) { consume(std::true_type{}, (uint32_t)u < ca, ra[u], acc[u], (uint32_t)i * RED_UNROLL + u); if constexpr (EPV == 8) __builtin_amdgcn_sched_barrier(0); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < RED_UNROLL; ++u) ra[u] = rb[u];
        }
    }
    if (ARG) {
        // the streamed vectors' winner of this lane as (key, index), then against the head / tail elements
        if (s_code != ~0u) {
            uint32_t k, code, pos, vb;
            if (s_nan != ~0u) { k = 0xFFFFFFFFu; code = s_nan / EPV; pos = s_nan % EPV; vb = s_nan_bits; }
            else {
                float w[EPV];
                RI::unpack(s_raw, w);
                pos = 0; vb = __float_as_uint(s_best);
#pragma unroll
                for (int c = EPV - 1; c >= 0; --c)
                    if (w[c] == s_best) { pos = (uint32_t)c; vb = __float_as_uint(w[c]); }       // first element at the extremum (-0 == +0)
                k = arg_key<AOP>(s_best);
                code = s_code;
            }
            const uint64_t vidx = item_base(code / RED_UNROLL) + (uint64_t)(code % RED_UNROLL) * RED_BLOCK;
            arg_combine3(best_key, best_idx, best_bits, k, vidx * EPV + head_n + pos, vb);
        }
    }
    RED_STAMP(2);

    // ragged tail (< one row): the last workgroup, guarded scalar loads
    const uint64_t tail_base = rows * ROW;
    if (tail_base < n && blockIdx.x == G - 1) {
        for (uint64_t i = tail_base + tid; i < n; i += RED_BLOCK) {
            const float v = RI::widen(in[i]);
            if (SUM) tail_acc = V::apply(tail_acc, v);
            if (NANS) nan_seen |= (v != v);
            if (ARG) {
                const uint32_t k = arg_key<AOP>(v);
                if (k > best_key) { best_key = k; best_idx = i + head_n; best_bits = __float_as_uint(v); }
            }
        }
    }

    // all LDS scratch in one object (hand-off flag included)
    __shared__ struct { float sum[RED_BLOCK / 64]; uint32_t key[RED_BLOCK / 64]; uint64_t idx[RED_BLOCK / 64]; uint32_t bits[RED_BLOCK / 64]; uint32_t last; } sh;
    const uint32_t lane = tid & 63u, wave = tid >> 6;

    if (SUM) {
        // fixed tree: slots pairwise, then the 4 vector components, then tail, then lanes
        f32x4 a = V::apply(V::apply(acc[0], acc[1]), V::apply(acc[2], acc[3]));
        f32x4 b = V::apply(V::apply(acc[4], acc[5]), V::apply(acc[6], acc[7]));
        f32x4 s = V::apply(a, b);
        float lane_sum = V::apply(V::apply(V::apply(s[0], s[1]), V::apply(s[2], s[3])), tail_acc);
        lane_sum = wave_fold<VOP == VOP_NONE ? VOP_SUM : VOP>(lane_sum);
        if (lane == 0) sh.sum[wave] = lane_sum;
    }
    if constexpr (NANS) {            // the NaN flag rides in the record's key word (free: no index operation in this pass)
        const bool any = __any(nan_seen);
        if (lane == 0) sh.key[wave] = any ? 1u : 0u;
    }
    if (ARG) {
        wave_argmax3(best_key, best_idx, best_bits);
        if (lane == 0) { sh.key[wave] = best_key; sh.idx[wave] = best_idx; sh.bits[wave] = best_bits; }
    }
    __syncthreads();
    if (tid == 0) {
        float rs = 0.f; uint32_t rk = 0u; uint64_t ri = ~0ull;
        if (SUM) rs = V::apply(V::apply(sh.sum[0], sh.sum[1]), V::apply(sh.sum[2], sh.sum[3]));
        if (NANS) rk = sh.key[0] | sh.key[1] | sh.key[2] | sh.key[3];
        if (ARG) {
            // the record carries the winner's VALUE BITS in its key word (the fold below re-derives the key): the last
            // workgroup then reports the element without a dependent load behind the fold (1.7 us of a 26 us pass)
            uint32_t rb = sh.bits[0];
            rk = sh.key[0]; ri = sh.idx[0];
#pragma unroll
            for (int w = 1; w < RED_BLOCK / 64; ++w) arg_combine3(rk, ri, rb, sh.key[w], sh.idx[w], sh.bits[w]);
            rk = rb;
        }
        if (poll) {                                                 // one 16-byte store with the valid bit, nothing to wait for
            poll_record_store(reinterpret_cast<red_record *>(ticket + RED_POLL_WORD0) + blockIdx.x, rs, rk, ri);
            RED_STAMP(3);
            sh.last = blockIdx.x == G - 1 ? 1u : 0u;
            RED_STAMP(4);
        } else {
            record_store(records + blockIdx.x, rs, rk, ri);        // write-through (sc1) 8-byte stores
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // ... drained before the ticket
            RED_STAMP(3);
            sh.last = arrive_is_last(ticket, G) ? 1u : 0u;
            RED_STAMP(4);
        }
    }
    __syncthreads();
    if (!sh.last) return;

    // ---- the last workgroup folds the G records in index order (thread t owns t, t+256, ...) ----
    float facc = V::identity();
    uint32_t key = 0u, bits = 0u;
    uint64_t idx = ~0ull;
    for (uint32_t gi = tid; gi < G; gi += RED_BLOCK) {
        // polling: waits for the record's valid bit (and clears it); tickets: every record landed before the last ticket was taken
        const red_record r = poll ? poll_record_take(reinterpret_cast<red_record *>(ticket + RED_POLL_WORD0) + gi)
                                  : record_load(records + gi);     // sc1 loads: served by L2, never a stale L1 line
        if (SUM) facc = V::apply(facc, r.sum);
        if (NANS) key |= r.key;
        if (ARG) arg_combine3(key, idx, bits, r.idx == ~0ull ? 0u : arg_key<AOP>(__uint_as_float(r.key)), r.idx, r.key);
    }
    __syncthreads();                                                // sh.* is reused below
    if (SUM) { facc = wave_fold<VOP == VOP_NONE ? VOP_SUM : VOP>(facc); if (lane == 0) sh.sum[wave] = facc; }
    if constexpr (NANS) { const bool any = __any(key != 0u); if (lane == 0) sh.key[wave] = any ? 1u : 0u; }
    if (ARG) { wave_argmax3(key, idx, bits); if (lane == 0) { sh.key[wave] = key; sh.idx[wave] = idx; sh.bits[wave] = bits; } }
    __syncthreads();
    if (tid == 0) {
        typedef __attribute__((address_space(1))) unsigned int gu32;
        if (!poll) __hip_atomic_store((gu32 *)ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next call
        if (SUM && out_sum) {
            float total = V::apply(V::apply(sh.sum[0], sh.sum[1]), V::apply(sh.sum[2], sh.sum[3]));
            if (NANS && (sh.key[0] | sh.key[1] | sh.key[2] | sh.key[3])) total = __uint_as_float(0x7FC00000u);
            if (VOP == VOP_SUM && mean_div != 0.f) total = total / mean_div;
            *out_sum = total;
        }
        if (ARG) {
            uint32_t k = sh.key[0], vb = sh.bits[0]; uint64_t ix = sh.idx[0];
#pragma unroll
            for (int w = 1; w < RED_BLOCK / 64; ++w) arg_combine3(k, ix, vb, sh.key[w], sh.idx[w], sh.bits[w]);
            if (n_total == 0) {
                if (out_idx) *out_idx = 0;
                if (out_val) *out_val = -NAN_LEADS;          // the identity: -inf (argmax) / +inf (argmin)
            } else {
                if (out_idx) *out_idx = ix;
                if (out_val) *out_val = __uint_as_float(vb);  // the winning element, bit for bit (it travelled with its index)
            }
        }
        RED_STAMP(5);
    }
}

#ifdef RED_TRACE
extern "C" __attribute__((visibility("default"))) int mi355_dev_red_trace(unsigned long long *host_out, int clear)
{
    const int rc = (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(red_trace_buf), sizeof(unsigned long long) * 4096 * 8);
    if (clear) { static unsigned long long z[4096 * 8]; (void)hipMemcpyToSymbol(HIP_SYMBOL(red_trace_buf), z, sizeof(z)); }
    return rc;
}
#endif

uint32_t pick_grid(const mi355_ctx *ctx, uint64_t n, uint64_t tile_elems = RED_TILE)
{
    const uint64_t tiles = (n + tile_elems - 1) / tile_elems;
    static const int per_cu = [] { const char *e = getenv("MI355_REDUCE_WG_PER_CU"); int v = e ? atoi(e) : 1; return v > 0 ? v : 1; }();
    const uint64_t cap = std::min<uint64_t>((uint64_t)ctx->props.num_streaming_multiprocessors * per_cu, RED_MAX_GRID);
    return (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(tiles, cap));
}

template <int VOP, int AOP, int DT>
int32_t run_reduce_t(mi355_ctx *ctx, mi355_stream stream, const void *in_v, uint64_t n, float *out_sum,
                     float *out_val, uint64_t *out_idx, void *workspace, uint64_t workspace_bytes, const char *what, float mean_div)
{
    typedef typename red_in<DT>::elem elem;
    constexpr uint32_t ESZ = sizeof(elem);
    const elem *in = static_cast<const elem *>(in_v);
    if (n && !in) return fail(ctx, MI355_E_INVALID_ARGUMENT, "%s: input is NULL", what);
    if ((reinterpret_cast<uintptr_t>(in) & (ESZ - 1)) != 0)
        return fail(ctx, MI355_E_INVALID_ARGUMENT, "%s: input must be aligned to its element size", what);
    uint64_t need = 0;
    mi355_reduce_workspace_bytes(ctx, n, &need);
    if (!workspace || workspace_bytes < need)
        return fail(ctx, MI355_E_INVALID_ARGUMENT, "%s: workspace too small (%llu < %llu bytes)", what,
                    (unsigned long long)workspace_bytes, (unsigned long long)need);
    if ((reinterpret_cast<uintptr_t>(workspace) & 15u) != 0)
        return fail(ctx, MI355_E_INVALID_ARGUMENT, "%s: workspace must be 16-byte aligned", what);
    hipStream_t s = stream_of(ctx, stream);
    // peel a misaligned head so the body is 16-byte aligned
    uint32_t head_n = (uint32_t)(((16u - (reinterpret_cast<uintptr_t>(in) & 15u)) & 15u) / ESZ);
    if (head_n > n) head_n = (uint32_t)n;
    const elem *body = in + head_n;
    const uint64_t body_n = n - head_n;
    const uint32_t G = pick_grid(ctx, body_n, (uint64_t)RED_BLOCK * RED_UNROLL * red_in<DT>::EPV);
    red_record *records = static_cast<red_record *>(workspace);
    unsigned int *ticket = nullptr;
    const int32_t trc = ticket_for_stream(ctx, s, &ticket);
    if (trc != MI355_OK) return trc;
    const uint64_t rounds = body_n / ((uint64_t)RED_BLOCK * red_in<DT>::EPV) / ((uint64_t)G * RED_UNROLL);
    // hand-off by polled records (library scratch) on grids that fit the slot, by tickets + the caller's workspace beyond (MI355_REDUCE_POLL=0: dev)
    static const bool poll_ok = [] { const char *e = getenv("MI355_REDUCE_POLL"); return !e || atoi(e) != 0; }();
    const uint32_t poll = (poll_ok && G <= RED_POLL_MAX) ? 1u : 0u;
    hipLaunchKernelGGL((reduce_kernel<VOP, AOP, DT>), dim3(G), dim3(RED_BLOCK), 0, s, in, head_n, body, body_n, records,
                       ticket, n, out_sum, out_val, out_idx, mean_div, rounds, poll);
    if (hipPeekAtLastError() != hipSuccess) ctx->tickets_dirty = true;   // a refused launch never resets its ticket
    check_launch(ctx, what);
    return MI355_OK;
}

template <int VOP, int AOP>
int32_t run_reduce(mi355_ctx *ctx, mi355_stream stream, const void *in, int32_t dtype, uint64_t n, float *out_sum,
                   float *out_val, uint64_t *out_idx, void *workspace, uint64_t workspace_bytes, const char *what, float mean_div = 0.f)
{
    MI355_REQUIRE_CTX(ctx);
    switch (dtype) {
    case MI355_DTYPE_F32: return run_reduce_t<VOP, AOP, MI355_DTYPE_F32>(ctx, stream, in, n, out_sum, out_val, out_idx, workspace, workspace_bytes, what, mean_div);
    case MI355_DTYPE_BF16: return run_reduce_t<VOP, AOP, MI355_DTYPE_BF16>(ctx, stream, in, n, out_sum, out_val, out_idx, workspace, workspace_bytes, what, mean_div);
    case MI355_DTYPE_F16: return run_reduce_t<VOP, AOP, MI355_DTYPE_F16>(ctx, stream, in, n, out_sum, out_val, out_idx, workspace, workspace_bytes, what, mean_div);
    default: return fail(ctx, MI355_E_UNSUPPORTED, "%s: input dtype %d (f32, bf16 or f16)", what, dtype);
    }
}

// ---- last-axis reductions --------------------------------------------------------------------
// One wave per row (THREADS=64) or one workgroup per row (THREADS=256/1024).  Rows are
// independent; the row is streamed with 16-B loads when its base and stride allow it.
// OP: the public operation code -- MI355_REDUCE_SUM / MEAN / MAX / MIN / PROD write f32 values, ARGMAX / ARGMIN u32 indices.
template <int OP> struct axis_op {
    static constexpr bool ARG = OP == MI355_REDUCE_ARGMAX || OP == MI355_REDUCE_ARGMIN;
    static constexpr int AOP = OP == MI355_REDUCE_ARGMIN ? AOP_MIN : AOP_MAX;
    static constexpr int VOP = (OP == MI355_REDUCE_MAX || OP == MI355_REDUCE_MIN || OP == MI355_REDUCE_PROD) ? OP : VOP_SUM;
    typedef vop<VOP> V;
    // the value written for a reduced axis of `count` elements whose fold is `t` (`nan`: MAX / MIN saw a NaN)
    static __device__ __forceinline__ float finish(float t, bool nan, uint64_t count)
    {
        if (V::TRACKS_NAN && nan) return __uint_as_float(0x7FC00000u);
        if (OP == MI355_REDUCE_MEAN) return t / (float)count;
        return t;
    }
};

template <int THREADS, int OP, int DT = MI355_DTYPE_F32>
__global__ void __launch_bounds__(THREADS)
reduce_rows(const typename red_in<DT>::elem *__restrict__ in, float *__restrict__ out_sum, uint32_t *__restrict__ out_idx, uint64_t rows,
            uint64_t cols_all, uint64_t row_stride, int vec_ok, uint32_t segs = 1, uint64_t seg_len = 0, float *__restrict__ part_val = nullptr,
            uint32_t *__restrict__ part_key = nullptr, uint64_t *__restrict__ part_idx = nullptr)
{
    // (late round 6) segs > 1: few, long rows -- every row is cut into `segs` spans of `seg_len` elements (the last one shorter), a workgroup folds one span and leaves the
    // raw fold (value + NaN flag, or key + index in the row) in library scratch; fold_row_segments finishes the rows.  13 x 992618 f32 ran on 13 workgroups: 270 us for 52 MB.
    typedef axis_op<OP> O;
    typedef typename O::V V;
    constexpr bool ARG = O::ARG;
    typedef red_in<DT> RI;
    constexpr int EPV = RI::EPV;
    constexpr int WAVES = THREADS / 64;
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    __shared__ float s_sum[WAVES];
    __shared__ uint32_t s_key[WAVES];
    __shared__ uint64_t s_idx[WAVES];
    constexpr bool SEG = THREADS == 256;                  // (only the 256-thread form is launched with spans: the others keep their per-row cost)
    const uint64_t vrows = SEG ? rows * segs : rows;
    for (uint64_t vr = blockIdx.x; vr < vrows; vr += gridDim.x) {
        const uint64_t row = (SEG && segs > 1) ? vr / segs : vr, first = (SEG && segs > 1) ? (vr % segs) * seg_len : 0;
        const uint64_t cols = (SEG && segs > 1) ? (cols_all - first < seg_len ? cols_all - first : seg_len) : cols_all;      // (the host leaves no span empty)
        const typename RI::elem *__restrict__ p = in + row * row_stride + first;
        auto put_val = [&](float t, bool nan) {
            if ((SEG && segs > 1)) { part_val[vr] = t; part_key[vr] = nan ? 1u : 0u; } else out_sum[row] = O::finish(t, nan, cols);
        };
        auto put_idx = [&](uint32_t k, uint64_t ix) {
            if ((SEG && segs > 1)) { part_key[vr] = k; part_idx[vr] = ix + first; } else out_idx[row] = cols ? (uint32_t)ix : 0u;
        };
        float a0 = V::identity(), a1 = V::identity(), a2 = V::identity(), a3 = V::identity();
        bool nan_seen = false;
        uint32_t key = 0u; uint64_t idx = ~0ull;
        // index operations: the value behind `key` -- a 16-byte vector only matters if it holds a NaN or something beyond it (the
        // fast reject of reduce_kernel: the exact key update then runs on a few percent of the vectors)
        constexpr bool AMIN = O::AOP == AOP_MIN;
        constexpr float NAN_LEADS = AMIN ? -__builtin_inff() : __builtin_inff();
        typedef vop<AMIN ? VOP_MIN : VOP_MAX> X;
        float best_val = -NAN_LEADS;
        // (late round 6: the 16-byte body starts at the ROW's own first aligned element -- until then one flag covered the launch, and a row length that is not a
        //  multiple of the vector (or a base off the 16-byte grid) sent every row through the scalar loop: 1404 x 133719 bf16 188 us = 0.25 of HBM, 635518 x 250 156 us)
        uint64_t head = 0, done = 0;           // (a row left to the scalar loop: no head, nothing done -- the plain increasing walk below)
        const uint64_t to_grid = ((16u - (uint32_t)(reinterpret_cast<uintptr_t>(p) & 15u)) & 15u) / (uint32_t)sizeof(typename RI::elem);
        // (a short row off the grid stays on the scalar loop as before: under 2 KiB the peel costs more than the vectors save -- 48382 x 147 bf16 8.1 us -> 14.9 with it --
        //  and rows of such a matrix that happen to start on the grid share their last cache line with the next row: non-temporal vectors fetch it twice -- 199410 x 316 bf16 46 -> 69 us)
        if ((vec_ok & 1) && to_grid <= cols && ((vec_ok & 2) || cols * sizeof(typename RI::elem) >= 2048)) {
            head = to_grid;
            const u32x4r *__restrict__ vp = reinterpret_cast<const u32x4r *>(p + head);
            const uint64_t nv = (cols - head) / EPV;
            constexpr int UL = 4;                                   // 16-byte loads in flight per lane (the data is read once: non-temporal)
            for (uint64_t i0 = tid; i0 < nv; i0 += (uint64_t)THREADS * UL) {
                u32x4r raw[UL];
#pragma unroll
                for (int u = 0; u < UL; ++u) {
                    const uint64_t i = i0 + (uint64_t)u * THREADS;
                    if (i < nv) raw[u] = __builtin_nontemporal_load(vp + i);
                }
#pragma unroll
                for (int u = 0; u < UL; ++u) {
                    const uint64_t i = i0 + (uint64_t)u * THREADS;
                    if (i >= nv) break;
                    float v[EPV];
                    RI::unpack(raw[u], v);
                    if (!ARG) {
                        a0 = V::apply(a0, v[0]); a1 = V::apply(a1, v[1]); a2 = V::apply(a2, v[2]); a3 = V::apply(a3, v[3]);
                        if constexpr (EPV == 8) { a0 = V::apply(a0, v[4]); a1 = V::apply(a1, v[5]); a2 = V::apply(a2, v[6]); a3 = V::apply(a3, v[7]); }
                        if constexpr (V::TRACKS_NAN) {
                            nan_seen |= __builtin_isunordered(v[0], v[1]) | __builtin_isunordered(v[2], v[3]);
                            if constexpr (EPV == 8) nan_seen |= __builtin_isunordered(v[4], v[5]) | __builtin_isunordered(v[6], v[7]);
                        }
                    } else {
                        float m4 = X::apply(X::apply(v[0], v[1]), X::apply(v[2], v[3]));
                        bool has_nan = __builtin_isunordered(v[0], v[1]) | __builtin_isunordered(v[2], v[3]);
                        if constexpr (EPV == 8) {
                            m4 = X::apply(m4, X::apply(X::apply(v[4], v[5]), X::apply(v[6], v[7])));
                            has_nan |= __builtin_isunordered(v[4], v[5]) | __builtin_isunordered(v[6], v[7]);
                        }
                        if ((AMIN ? (m4 < best_val) : (m4 > best_val)) | has_nan | (key == 0u)) {
#pragma unroll
                            for (int c = 0; c < EPV; ++c) {
                                const uint32_t k = arg_key<O::AOP>(v[c]);
                                if (k > key) { key = k; idx = head + i * EPV + c; best_val = (k == 0xFFFFFFFFu) ? NAN_LEADS : v[c]; }
                            }
                        }
                    }
                }
            }
            done = head + nv * EPV;
        }
        // the elements before the first and after the last whole vector (every element when the base is off its element grid).  A lane meets the head's
        // indices after its body's: equal keys keep the lower index explicitly
        for (uint64_t i = tid; i < head; i += THREADS) {
            const float v = RI::widen(p[i]);
            if (!ARG) { a0 = V::apply(a0, v); if (V::TRACKS_NAN) nan_seen |= (v != v); }
            else { const uint32_t k = arg_key<O::AOP>(v); if (k > key || (k == key && i < idx)) { key = k; idx = i; } }
        }
        for (uint64_t i = done + tid; i < cols; i += THREADS) {
            const float v = RI::widen(p[i]);
            if (!ARG) { a0 = V::apply(a0, v); if (V::TRACKS_NAN) nan_seen |= (v != v); }
            else { const uint32_t k = arg_key<O::AOP>(v); if (k > key) { key = k; idx = i; } }
        }
        if (!ARG) {
            float s = wave_fold<O::VOP>(V::apply(V::apply(a0, a1), V::apply(a2, a3)));
            const bool wave_nan = V::TRACKS_NAN ? (bool)__any(nan_seen) : false;
            if (WAVES == 1) { if (lane == 0) put_val(s, wave_nan); }
            else {
                if (lane == 0) { s_sum[wave] = s; s_key[wave] = wave_nan ? 1u : 0u; }
                __syncthreads();
                if (tid == 0) {
                    float t = V::identity(); uint32_t nn = 0u;
                    for (int w = 0; w < WAVES; ++w) { t = V::apply(t, s_sum[w]); nn |= s_key[w]; }
                    put_val(t, nn != 0u);
                }
                __syncthreads();
            }
        } else {
            wave_argmax(key, idx);
            if (WAVES == 1) { if (lane == 0) put_idx(key, idx); }
            else {
                if (lane == 0) { s_key[wave] = key; s_idx[wave] = idx; }
                __syncthreads();
                if (tid == 0) {
                    uint32_t k = s_key[0]; uint64_t ix = s_idx[0];
                    for (int w = 1; w < WAVES; ++w) arg_combine(k, ix, s_key[w], s_idx[w]);
                    put_idx(k, ix);
                }
                __syncthreads();
            }
        }
    }
}

// ---- reductions over a NON-last axis: in is [outer][reduce][inner] (contiguous), out is [outer][inner] ------------
// Roofline: HBM, the input read once.  Until round 4 one thread walked the whole axis for one `inner` position with one load
// in flight: 0.3-3.4 TB/s (8192 x 8192 over axis 0 = 256 MiB: 292 us, 0.9 TB/s; 4 x 65536 x 1024 over axis 1: 3.4 ms, 0.3 TB/s --
// profiles/r04_axis_probe_before.txt).  Now a workgroup owns a TILE: TX threads along `inner` (16 bytes each when rows allow,
// so a wave reads whole lines) x TY = 256 / TX threads along the reduced axis, eight rows in flight per thread; the TY partial
// vectors meet in LDS in thread order.  When outer x inner alone cannot fill the chip the reduced axis is cut into `chunks`
// row ranges, every workgroup writes its partial to library scratch and a second launch of the SAME kernel reduces the
// [outer][chunks][inner] partials (value operations: max / min carry NaN as a NaN partial, mean divides at the very end; index
// operations: (key, row index) pairs, folded with the lowest-index rule).  The tree is a function of the shape only:
// deterministic.
constexpr int SCRATCH_REDUCE_AXIS = 6;      // library scratch kind (gemm_common.hpp lists 0-5)

struct axis_args {
    const void *in;                 // stage 1: elements of DT; stage 2 of a value operation: f32 partials
    const uint32_t *in_key, *in_idx;   // stage 2 of an index operation: partial (key, row index) pairs
    float *out_val;                 // final f32 values, or the partial values when `partial`
    uint32_t *out_key, *out_idx;    // final u32 indices (out_idx), or partial pairs
    uint64_t outer, red, inner;
    uint64_t rows_per_chunk;        // rows of the reduced axis per workgroup (multiple of TY)
    uint32_t chunks, inner_blocks, log2_tx, partial;
    uint32_t log2_ty;               // TY = threads along the reduced axis; the remaining 256 / (TX x TY) thread groups take one `outer` slice each
    uint64_t outer_blocks;          // workgroups along `outer` (OY slices each)
    float mean_div;                 // != 0: the final value is sum / mean_div
};

template <int OP, int DT, bool VECTOR, bool PAIRS>
__global__ void __launch_bounds__(256)
reduce_axis_tiled(axis_args a)
{
    typedef axis_op<OP> O;
    typedef typename O::V V;
    constexpr bool ARG = O::ARG;
    typedef red_in<DT> RI;
    constexpr int W = VECTOR ? RI::EPV : 1;                 // elements per thread along `inner`
    constexpr int U = 8;                                    // rows in flight per thread
    static_assert(!PAIRS || (ARG && !VECTOR), "pair input: the second stage of an index operation, one position per thread");
    __shared__ uint32_t s_a[256 * W], s_b[256 * W];         // per thread: W values (or keys) and W NaN flags (or row indices)

    // thread -> (outer slice oy of this workgroup, row lane ty, column lane tx): a short axis under a narrow `inner` leaves most of a
    // 256-thread tile without rows, so the spare threads take further `outer` slices (1 048 576 x 16 x 4 over axis 1: one workgroup
    // per 128 slices = 32 KiB contiguous, instead of one workgroup per slice)
    const uint32_t tid = threadIdx.x, TX = 1u << a.log2_tx, TY = 1u << a.log2_ty, OY = 256u >> (a.log2_tx + a.log2_ty);
    const uint32_t tx = tid & (TX - 1), ty = (tid >> a.log2_tx) & (TY - 1), oy = tid >> (a.log2_tx + a.log2_ty);
    // linear workgroup id -> (outer block ob, chunk c, inner block ib), inner block fastest
    const uint64_t wg = blockIdx.x;
    const uint32_t ib = (uint32_t)(wg % a.inner_blocks);
    const uint64_t rest = wg / a.inner_blocks;
    const uint32_t c = (uint32_t)(rest % a.chunks);
    const uint64_t o = (rest / a.chunks) * OY + oy;
    const uint64_t i0 = ((uint64_t)ib * TX + tx) * W;
    const bool live = i0 < a.inner && o < a.outer;
    const uint64_t r0 = (uint64_t)c * a.rows_per_chunk, r1 = min(a.red, r0 + a.rows_per_chunk);

    float acc[W];
    uint32_t key[W], idx[W];
    uint32_t nanbits = 0u;           // max / min: bit e = column e of this thread has seen a NaN (one VGPR; W booleans lived in SGPR pairs and spilled)
    // index operations: acc[e] doubles as the value behind key[e] (nothing seen yet: NaN; a NaN in the lead: +inf / argmin -inf) -- a vector is
    // keyed only if some element is beyond its column's value or a NaN; eight keys per 16 bytes of bf16 would otherwise make the kernel VALU-bound
    constexpr bool AMIN = O::AOP == AOP_MIN;
    constexpr float NAN_LEADS = AMIN ? -__builtin_inff() : __builtin_inff();
#pragma unroll
    for (int e = 0; e < W; ++e) { acc[e] = ARG ? __builtin_nanf("") : V::identity(); key[e] = 0u; idx[e] = 0u; }

    if (live) {
        const uint64_t base = (o * a.red) * a.inner + i0;
        // one row's W elements (or one partial pair) into the running state
        auto consume = [&](const float (&x)[W], uint32_t row) {
            if constexpr (ARG) {
                // One test per VECTOR (a branch per element cost more than the keys it saved): "not (x <= running extremum)" is true for
                // an element beyond it, for a NaN, and while the column has seen nothing (acc starts as NaN); once a NaN leads acc is
                // +inf (argmin: -inf) and only another NaN passes, to lose on the key.
                bool cand = false;
#pragma unroll
                for (int e = 0; e < W; ++e) cand |= AMIN ? !(x[e] >= acc[e]) : !(x[e] <= acc[e]);
                if (cand) {
#pragma unroll
                    for (int e = 0; e < W; ++e) {
                        const uint32_t k = arg_key<O::AOP>(x[e]);
                        const bool take = k > key[e];                    // rows ascend per thread: the first extremum stays
                        key[e] = take ? k : key[e];
                        idx[e] = take ? row : idx[e];
                        acc[e] = take ? ((k == 0xFFFFFFFFu) ? NAN_LEADS : x[e]) : acc[e];
                    }
                }
            } else {
#pragma unroll
                for (int e = 0; e < W; ++e) {
                    acc[e] = V::apply(acc[e], x[e]);
                    if (V::TRACKS_NAN) nanbits |= (x[e] != x[e]) ? (1u << e) : 0u;
                }
            }
        };
        auto load_raw = [&](uint64_t rr, u32x4r &raw, typename RI::elem &one, uint32_t &pk, uint32_t &pi) {
            const uint64_t at = base + rr * a.inner;
            if constexpr (PAIRS) { pk = a.in_key[at]; pi = a.in_idx[at]; }
            else if constexpr (VECTOR) raw = __builtin_nontemporal_load(reinterpret_cast<const u32x4r *>(static_cast<const typename RI::elem *>(a.in) + at));
            else one = static_cast<const typename RI::elem *>(a.in)[at];
        };
        auto take_raw = [&](uint64_t rr, const u32x4r &raw, typename RI::elem one, uint32_t pk, uint32_t pi) {
            if constexpr (PAIRS) arg_combine_u32(key[0], idx[0], pk, pi);
            else {
                float x[W];
                if constexpr (VECTOR) RI::unpack(raw, x); else x[0] = RI::widen(one);
                consume(x, (uint32_t)rr);
            }
        };
        uint64_t r = r0 + ty;
        // full groups of U rows: all U loads first, RAW (a 16-bit vector is unpacked only when it is consumed -- unpacking inside the load
        // loop made every load wait for the one before it: one load in flight, 2.3 TB/s on bf16 where f32 ran 6.5), no bounds tests
        for (; r + (uint64_t)(U - 1) * TY < r1; r += (uint64_t)TY * U) {
            u32x4r raw[U];
            typename RI::elem one[U];
            uint32_t pk[U], pi[U];
#pragma unroll
            for (int u = 0; u < U; ++u) load_raw(r + (uint64_t)u * TY, raw[u], one[u], pk[u], pi[u]);
#pragma unroll
            for (int u = 0; u < U; ++u) take_raw(r + (uint64_t)u * TY, raw[u], one[u], pk[u], pi[u]);
        }
        for (; r < r1; r += TY) {                                 // the ragged rest, one row at a time
            u32x4r raw; typename RI::elem one; uint32_t pk, pi;
            load_raw(r, raw, one, pk, pi);
            take_raw(r, raw, one, pk, pi);
        }
    }
    // ---- the TY partial vectors of a column meet in LDS, folded by the ty == 0 thread in ty order ----------------------------
    if (TY > 1) {
#pragma unroll
        for (int e = 0; e < W; ++e) {
            s_a[tid * W + e] = ARG ? key[e] : __float_as_uint(acc[e]);
            s_b[tid * W + e] = ARG ? idx[e] : ((nanbits >> e) & 1u);
        }
        __syncthreads();
        if constexpr (ARG) {
            // (end of round 6) index operations: a tree over ty instead of the ty == 0 thread's walk -- arg_combine is a total order (larger key, then lower index), so the
            // tree gives the walk's answer bit for bit; the walk's 255 dependent compare-and-select steps per column were most of the kernel under a narrow `inner`
            // (744 x 2651 x 4 f32: argmax 83 us where the sum takes 12.5).  Value operations keep the walk: their result depends on the order.
            for (uint32_t half = TY >> 1; half >= 1; half >>= 1) {
                if (ty < half) {
                    const uint32_t other = (oy << (a.log2_tx + a.log2_ty)) + ((ty + half) << a.log2_tx) + tx;
#pragma unroll
                    for (int e = 0; e < W; ++e) arg_combine_u32(key[e], idx[e], s_a[other * W + e], s_b[other * W + e]);
                }
                __syncthreads();
                if (ty < half) {
#pragma unroll
                    for (int e = 0; e < W; ++e) { s_a[tid * W + e] = key[e]; s_b[tid * W + e] = idx[e]; }
                }
                __syncthreads();
            }
        } else if (ty == 0) {
            for (uint32_t t = 1; t < TY; ++t) {
                const uint32_t other = (oy << (a.log2_tx + a.log2_ty)) + (t << a.log2_tx) + tx;
#pragma unroll
                for (int e = 0; e < W; ++e) {
                    if constexpr (ARG) arg_combine_u32(key[e], idx[e], s_a[other * W + e], s_b[other * W + e]);
                    else { acc[e] = V::apply(acc[e], __uint_as_float(s_a[other * W + e])); nanbits |= s_b[other * W + e] << e; }
                }
            }
        }
    }
    if (ty != 0 || !live) return;
    const uint64_t out_at = a.partial ? (o * a.chunks + c) * a.inner + i0 : o * a.inner + i0;
#pragma unroll
    for (int e = 0; e < W; ++e) {
        if constexpr (ARG) {
            if (a.partial) { a.out_key[out_at + e] = key[e]; a.out_idx[out_at + e] = idx[e]; }
            else a.out_idx[out_at + e] = idx[e];
        } else {
            float t = acc[e];
            if (V::TRACKS_NAN && ((nanbits >> e) & 1u)) t = __uint_as_float(0x7FC00000u);
            if (!a.partial && a.mean_div != 0.f) t = t / a.mean_div;
            a.out_val[out_at + e] = t;
        }
    }
}

// thread / tile geometry of one stage: TX threads along inner (W elements each), chunks of the reduced axis
struct axis_geom { uint32_t log2_tx, log2_ty, inner_blocks, chunks; uint64_t rows_per_chunk, outer_blocks; };
inline axis_geom axis_plan(uint64_t outer, uint64_t red, uint64_t inner, int w, uint64_t cus, bool allow_chunks)
{
    axis_geom g{};
    const uint64_t vecs = (inner + w - 1) / w;
    uint32_t l = 0;
    while (l < 8 && (1ull << l) < vecs) ++l;                 // TX = smallest power of two covering the row, at most 256
    g.log2_tx = l;
    const uint64_t tx = 1ull << l;
    // row lanes: as many as the tile has left, but no more than give every thread about eight rows (one unrolled trip)
    uint32_t lt = 0;
    while ((tx << (lt + 1)) <= 256 && (8ull << (lt + 1)) <= std::max<uint64_t>(red, 8)) ++lt;
    if (outer == 1) lt = 8 - l;                               // nothing else to give the spare threads to
    g.log2_ty = lt;
    const uint64_t ty = 1ull << lt, oy = 256 / (tx * ty);
    g.outer_blocks = (outer + oy - 1) / oy;
    g.inner_blocks = (uint32_t)((vecs + tx - 1) / tx);
    const uint64_t base = std::max<uint64_t>(1, g.outer_blocks * g.inner_blocks);
    uint64_t chunks = 1;
    // workgroups per CU the cut aims at: ONE measured best (1 / 2 / 3 / 4 / 6 / 8 / 12 per CU, f32 sums, median of 15 samples, two rounds:
    // 8192 x 8192 over axis 0 50 / 57 / 54 / 66 / 63 / 73 / 74 us, 4 x 65536 x 1024 over axis 1 165 / 169 / 179 / 183 / 200 / 217 / 240,
    // 16384^2 over axis 0 166 / 176 / 167 / 179 / 175 / 213 / 192 -- more chunks only add partials and a longer second stage;
    // profiles/r04_axis_wg_per_cu_sweep.txt)
    static const uint64_t per_cu = [] { const char *e = getenv("MI355_AXIS_WG_PER_CU"); const int v = e ? atoi(e) : 1; return (uint64_t)(v > 0 ? v : 1); }();
    if (allow_chunks && base < cus * per_cu) {
        chunks = std::min<uint64_t>((cus * per_cu + base - 1) / base, std::max<uint64_t>(1, red / (ty * 16)));   // >= two unrolled trips per workgroup
        chunks = std::min<uint64_t>(chunks, 1024);
    }
    uint64_t rows = (red + chunks - 1) / std::max<uint64_t>(chunks, 1);
    rows = std::max<uint64_t>(ty, (rows + ty - 1) / ty * ty);
    g.rows_per_chunk = rows;
    g.chunks = (uint32_t)std::max<uint64_t>(1, (red + rows - 1) / rows);
    return g;
}

template <int OP, int DT = MI355_DTYPE_F32>
int32_t run_mid(mi355_ctx *ctx, mi355_stream stream, const typename red_in<DT>::elem *in, float *out_sum, uint32_t *out_idx, uint64_t outer,
                uint64_t red, uint64_t inner, const char *what)
{
    constexpr bool ARG = axis_op<OP>::ARG;
    // the kernel folds SUM and divides at the very end for MEAN
    constexpr int KOP = OP == MI355_REDUCE_MEAN ? MI355_REDUCE_SUM : OP;
    MI355_REQUIRE_CTX(ctx);
    if (outer == 0 || inner == 0) return MI355_OK;
    if ((red && !in) || (!ARG && !out_sum) || (ARG && !out_idx)) return fail(ctx, MI355_E_INVALID_ARGUMENT, "%s: NULL pointer", what);
    if (ARG && red > 0xFFFFFFFFull) return fail(ctx, MI355_E_UNSUPPORTED, "%s: reduced axis exceeds the u32 index range", what);
    hipStream_t s = stream_of(ctx, stream);
    const uint64_t cus = ctx->props.num_streaming_multiprocessors;
    constexpr int EPV = red_in<DT>::EPV;
    const bool vec = (inner % EPV) == 0 && (reinterpret_cast<uintptr_t>(in) & 15u) == 0;
    const float mean_div = OP == MI355_REDUCE_MEAN ? (red ? (float)red : __builtin_nanf("")) : 0.f;      // (an empty axis has no mean: 0 / 0)
    axis_geom g = axis_plan(outer, red, inner, vec ? EPV : 1, cus, true);
    if (g.outer_blocks * (uint64_t)g.inner_blocks * g.chunks > 0x7FFFFFFFull) return fail(ctx, MI355_E_UNSUPPORTED, "%s: too many tiles", what);
    axis_args a{};
    a.in = in; a.outer = outer; a.red = red; a.inner = inner;
    auto set_geom = [](axis_args &x, const axis_geom &q) {
        x.log2_tx = q.log2_tx; x.log2_ty = q.log2_ty; x.inner_blocks = q.inner_blocks; x.chunks = q.chunks; x.rows_per_chunk = q.rows_per_chunk;
        x.outer_blocks = q.outer_blocks;
    };
    set_geom(a, g);
    void *scratch = nullptr;
    if (g.chunks > 1) {
        const size_t per = (size_t)outer * g.chunks * inner * 4;
        if (scratch_get(ctx, s, SCRATCH_REDUCE_AXIS, ARG ? 2 * per : per, &scratch) != MI355_OK) {     // no scratch: one workgroup per column block
            g = axis_plan(outer, red, inner, vec ? EPV : 1, cus, false);
            set_geom(a, g);
        }
    }
    const bool two = g.chunks > 1;
    a.partial = two ? 1u : 0u;
    a.mean_div = two ? 0.f : mean_div;
    if (two) {
        a.out_val = static_cast<float *>(scratch);
        a.out_key = static_cast<uint32_t *>(scratch);
        a.out_idx = static_cast<uint32_t *>(scratch) + (size_t)outer * g.chunks * inner;
    } else { a.out_val = out_sum; a.out_idx = out_idx; }
    const uint32_t grid1 = (uint32_t)(g.outer_blocks * (uint64_t)g.inner_blocks * g.chunks);
    if (vec) hipLaunchKernelGGL((reduce_axis_tiled<KOP, DT, true, false>), dim3(grid1), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((reduce_axis_tiled<KOP, DT, false, false>), dim3(grid1), dim3(256), 0, s, a);
    if (two) {
        // stage 2: the partials as [outer][chunks][inner], one workgroup per column block (chunks <= 1024 rows)
        const bool vec2 = !ARG && (inner % 4) == 0;
        const axis_geom g2 = axis_plan(outer, g.chunks, inner, vec2 ? 4 : 1, cus, false);
        axis_args b{};
        b.in = scratch; b.in_key = static_cast<const uint32_t *>(scratch); b.in_idx = static_cast<const uint32_t *>(scratch) + (size_t)outer * g.chunks * inner;
        b.outer = outer; b.red = g.chunks; b.inner = inner;
        set_geom(b, g2);
        b.chunks = 1;
        b.partial = 0; b.mean_div = mean_div; b.out_val = out_sum; b.out_idx = out_idx;
        const uint32_t grid2 = (uint32_t)(g2.outer_blocks * (uint64_t)g2.inner_blocks);
        if constexpr (ARG) hipLaunchKernelGGL((reduce_axis_tiled<KOP, MI355_DTYPE_F32, false, true>), dim3(grid2), dim3(256), 0, s, b);
        else if (vec2) hipLaunchKernelGGL((reduce_axis_tiled<KOP, MI355_DTYPE_F32, true, false>), dim3(grid2), dim3(256), 0, s, b);
        else hipLaunchKernelGGL((reduce_axis_tiled<KOP, MI355_DTYPE_F32, false, false>), dim3(grid2), dim3(256), 0, s, b);
    }
    check_launch(ctx, what);
    return MI355_OK;
}

// Rows of at most 512 elements (end of round 6): one wave per row has one or two loads per lane in flight and then folds -- a chain of load latency + butterfly per
// row, 635518 x 250 bf16 at 0.25 of HBM.  Here a wave takes FOUR rows at a time: all their elements are requested first (16 or 32 plain loads per lane -- any alignment, and
// neighbouring rows share cache lines), then each row is folded the usual way (the reference's plane_reduce butterfly; lowest index among equal keys).
template <int OP, int DT, int J>
__global__ void __launch_bounds__(64)
reduce_short_rows(const typename red_in<DT>::elem *__restrict__ in, float *__restrict__ out_sum, uint32_t *__restrict__ out_idx, uint64_t rows, uint32_t cols,
                  uint64_t row_stride)
{
    typedef axis_op<OP> O;
    typedef typename O::V V;
    typedef red_in<DT> RI;
    constexpr int R = 4;                     // (J: 64-element slices of a row, 4 up to 256 elements, 8 up to 512)
    const uint32_t lane = threadIdx.x;
    for (uint64_t r0 = (uint64_t)blockIdx.x * R; r0 < rows; r0 += (uint64_t)gridDim.x * R) {
        float v[R][J];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint64_t row = r0 + r < rows ? r0 + r : rows - 1;            // (a row past the end re-reads the last one; its result is not written)
            const typename RI::elem *__restrict__ p = in + row * row_stride;
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const uint32_t c = lane + 64u * j;
                v[r][j] = RI::widen(p[c < cols ? c : 0]);                      // (unconditional loads: all sixteen are in flight before the first use)
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (r0 + r >= rows) break;
            if constexpr (!O::ARG) {
                float a = V::identity();
                bool nan_seen = false;
#pragma unroll
                for (int j = 0; j < J; ++j)
                    if (lane + 64u * j < cols) { a = V::apply(a, v[r][j]); if (V::TRACKS_NAN) nan_seen |= (v[r][j] != v[r][j]); }
                const float s = wave_fold<O::VOP>(a);
                const bool wave_nan = V::TRACKS_NAN ? (bool)__any(nan_seen) : false;
                if (lane == 0) out_sum[r0 + r] = O::finish(s, wave_nan, cols);
            } else {
                uint32_t key = 0u; uint64_t idx = ~0ull;
#pragma unroll
                for (int j = 0; j < J; ++j) {
                    const uint32_t c = lane + 64u * j;
                    if (c < cols) { const uint32_t k = arg_key<O::AOP>(v[r][j]); if (k > key) { key = k; idx = c; } }
                }
                wave_argmax(key, idx);
                if (lane == 0) out_idx[r0 + r] = cols ? (uint32_t)idx : 0u;
            }
        }
    }
}

// the rows of a segmented reduce_rows launch: one thread per row walks its spans in order (at most 1024 of them)
template <int OP>
__global__ void __launch_bounds__(64)
fold_row_segments(const float *__restrict__ part_val, const uint32_t *__restrict__ part_key, const uint64_t *__restrict__ part_idx, uint32_t segs,
                  uint64_t rows, uint64_t cols, float *__restrict__ out_sum, uint32_t *__restrict__ out_idx)
{
    typedef axis_op<OP> O;
    typedef typename O::V V;
    const uint64_t row = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    if (row >= rows) return;
    const uint64_t at = row * segs;
    if constexpr (!O::ARG) {
        float t = V::identity(); uint32_t nn = 0u;
        for (uint32_t g = 0; g < segs; ++g) { t = V::apply(t, part_val[at + g]); nn |= part_key[at + g]; }
        out_sum[row] = O::finish(t, nn != 0u, cols);
    } else {
        uint32_t k = part_key[at]; uint64_t ix = part_idx[at];
        for (uint32_t g = 1; g < segs; ++g) arg_combine(k, ix, part_key[at + g], part_idx[at + g]);
        out_idx[row] = (uint32_t)ix;
    }
}

template <int OP, int DT = MI355_DTYPE_F32>
int32_t run_rows(mi355_ctx *ctx, mi355_stream stream, const typename red_in<DT>::elem *in, float *out_sum, uint32_t *out_idx,
                 uint64_t rows, uint64_t cols, uint64_t row_stride, const char *what)
{
    constexpr bool ARG = axis_op<OP>::ARG;
    MI355_REQUIRE_CTX(ctx);
    if (rows == 0) return MI355_OK;
    if ((cols && !in) || (!ARG && !out_sum) || (ARG && !out_idx))
        return fail(ctx, MI355_E_INVALID_ARGUMENT, "%s: NULL pointer", what);
    if (row_stride < cols && rows > 1)
        return fail(ctx, MI355_E_UNSUPPORTED_STRIDES, "%s: row stride %llu < cols %llu", what,
                    (unsigned long long)row_stride, (unsigned long long)cols);
    if (ARG && cols > 0xFFFFFFFFull)
        return fail(ctx, MI355_E_UNSUPPORTED, "%s: cols exceed u32 index range", what);
    hipStream_t s = stream_of(ctx, stream);
    // bit 0: the base is on its element grid (each row finds its own first 16-byte boundary); bit 1: every row starts on the 16-byte grid
    const int vec_ok = ((reinterpret_cast<uintptr_t>(in) & (sizeof(typename red_in<DT>::elem) - 1)) == 0 ? 1 : 0) |
                       (((reinterpret_cast<uintptr_t>(in) & 15u) == 0 && (row_stride % red_in<DT>::EPV) == 0) ? 2 : 0);
    const uint64_t cus = ctx->props.num_streaming_multiprocessors;
    const uint64_t cols32 = cols * sizeof(typename red_in<DT>::elem) / 4;        // row length in f32-equivalents (bytes / 4)
    // few, long rows (late round 6, the seeded roofline audit of the axis reductions: 13 x 992618 f32 270 us = 0.02 of HBM on 13 workgroups): spans of at least 32 KiB, about
    // eight workgroups of 256 threads per CU; not inside a capture window without scratch (the unsplit launch below is always valid)
    if (cols32 >= 32768 && rows < cus * 2) {
        uint64_t segs = std::min<uint64_t>({(cus * 8 + rows - 1) / rows, cols32 / 8192, 1024});
        if (segs >= 2) {
            constexpr uint64_t GRAIN = 1024;                                       // span length: whole 16-byte vectors for every lane of a wave
            const uint64_t seg_len = ((cols + segs - 1) / segs + GRAIN - 1) / GRAIN * GRAIN;
            segs = (cols + seg_len - 1) / seg_len;
            void *scratch = nullptr;
            const uint64_t vrows = rows * segs;
            if (segs >= 2 && scratch_get(ctx, s, SCRATCH_REDUCE_AXIS, (size_t)vrows * 16, &scratch) == MI355_OK) {
                uint64_t *part_idx = static_cast<uint64_t *>(scratch);
                float *part_val = reinterpret_cast<float *>(part_idx + vrows);
                uint32_t *part_key = reinterpret_cast<uint32_t *>(part_val + vrows);
                const uint32_t grid = (uint32_t)std::min<uint64_t>(vrows, cus * 8);
                hipLaunchKernelGGL((reduce_rows<256, OP, DT>), dim3(grid), dim3(256), 0, s, in, out_sum, out_idx, rows, cols, row_stride, vec_ok, (uint32_t)segs, seg_len,
                                   part_val, part_key, part_idx);
                hipLaunchKernelGGL((fold_row_segments<OP>), dim3((uint32_t)((rows + 63) / 64)), dim3(64), 0, s, part_val, part_key, part_idx, (uint32_t)segs, rows, cols,
                                   out_sum, out_idx);
                check_launch(ctx, what);
                return MI355_OK;
            }
        }
    }
    if (cols <= 512 && cols > 0 && rows >= 4) {
        const uint32_t grid = (uint32_t)std::min<uint64_t>((rows + 3) / 4, cus * 32);
        if (cols <= 256) hipLaunchKernelGGL((reduce_short_rows<OP, DT, 4>), dim3(grid), dim3(64), 0, s, in, out_sum, out_idx, rows, (uint32_t)cols, row_stride);
        else hipLaunchKernelGGL((reduce_short_rows<OP, DT, 8>), dim3(grid), dim3(64), 0, s, in, out_sum, out_idx, rows, (uint32_t)cols, row_stride);
    } else if (cols32 <= 2048) {
        const uint32_t grid = (uint32_t)std::min<uint64_t>(rows, cus * 32);
        hipLaunchKernelGGL((reduce_rows<64, OP, DT>), dim3(grid), dim3(64), 0, s, in, out_sum, out_idx, rows, cols,
                           row_stride, vec_ok);
    } else if (cols32 <= 65536) {
        const uint32_t grid = (uint32_t)std::min<uint64_t>(rows, cus * 8);
        hipLaunchKernelGGL((reduce_rows<256, OP, DT>), dim3(grid), dim3(256), 0, s, in, out_sum, out_idx, rows, cols,
                           row_stride, vec_ok);
    } else {
        const uint32_t grid = (uint32_t)std::min<uint64_t>(rows, cus * 2);
        hipLaunchKernelGGL((reduce_rows<1024, OP, DT>), dim3(grid), dim3(1024), 0, s, in, out_sum, out_idx, rows, cols,
                           row_stride, vec_ok);
    }
    check_launch(ctx, what);
    return MI355_OK;
}

// ---- plane ops: one 64-lane plane per 64 inputs ---------------------------------------------
// (round 4: four planes = four waves per workgroup, grid-striding over the planes -- one 64-thread workgroup per plane ran a
//  256 MiB input as a million workgroups)
__global__ void __launch_bounds__(256)
plane_reduce_kernel(const float *__restrict__ in, float *__restrict__ out, uint64_t n, uint32_t active, int op)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t planes = (n + 63) / 64;
  for (uint64_t pl = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6); pl < planes; pl += (uint64_t)gridDim.x * 4) {
    const uint64_t i = pl * 64 + lane;
    float v = (i < n) ? in[i] : 0.f;
    if (op >= 101 && op <= 104) {
        // plane_reduce_inclusive / exclusive (shared/plane.rs:72-97): Hillis-Steele with shuffle_up; sum (101 / 102, default 0)
        // or product (103 / 104, default 1: lower_unop!(ExclusiveFProdOp, plane_reduce_exclusive, OpMul, 1), shared/plane.rs:133-136)
        const bool mul = op >= 103, exclusive = (op == 102 || op == 104);
        float acc = v;
        for (uint32_t off = 1; off < active; off <<= 1) {
            const float up = __shfl_up(acc, off, 64);
            if ((lane & (active - 1)) >= off) acc = mul ? acc * up : acc + up;
        }
        if (exclusive) {
            const float prev = __shfl_up(acc, 1, 64);
            acc = ((lane & (active - 1)) == 0) ? (mul ? 1.f : 0.f) : prev;
        }
        v = acc;
    } else {
        // plane_reduce (shared/plane.rs:60-70): xor butterfly, offsets 1,2,4,.. < active
        for (uint32_t off = 1; off < active; off <<= 1) {
            const float o = __shfl_xor(v, off, 64);
            switch (op) {
            case MI355_REDUCE_SUM: v = v + o; break;
            case MI355_REDUCE_MAX: v = v > o ? v : o; break;
            case MI355_REDUCE_MIN: v = v < o ? v : o; break;
            default: v = v * o; break;  // MI355_REDUCE_PROD (or its round-1 code 100): product
            }
        }
    }
    if (i < n) out[i] = v;
  }
}

// ---- the remaining plane intrinsics (frontend/plane.rs:62-216, :388-440), one plane of `plane` lanes per block --------
// Launched with blockDim = plane (32 or 64), exactly as the reference's tests launch CubeDim::new_1d(plane_size) on a
// wave64 device: the lanes beyond `plane` do not exist, so __activemask / __shfl see what a CubeCL kernel would see.
// Lowering follows crates/cubecl-cpp/src/hip/plane.rs (:19-64: __shfl / __shfl_xor / __shfl_up / __shfl_down / __all / __any /
// __ballot) and shared/plane.rs:170-174 (elect = lowest lane of the active mask).
__global__ void __launch_bounds__(64)
plane_op_kernel(const float *__restrict__ in, void *__restrict__ out_raw, uint64_t n, int op, uint32_t arg)
{
    const uint32_t lane = threadIdx.x, plane = blockDim.x;
    const uint64_t i = (uint64_t)blockIdx.x * plane + lane;
    if (i >= n) return;                                       // a ragged last plane simply has fewer active lanes
    const float v = in[i];
    float *out = static_cast<float *>(out_raw);
    switch (op) {
    case MI355_PLANE_ALL: out[i] = __all(v != 0.f) ? 1.f : 0.f; break;
    case MI355_PLANE_ANY: out[i] = __any(v != 0.f) ? 1.f : 0.f; break;
    case MI355_PLANE_ELECT: out[i] = ((uint32_t)__builtin_ctzll(__ballot(1)) == lane) ? 1.f : 0.f; break;
    case MI355_PLANE_BROADCAST:
    // width = the plane: a source lane outside it leaves the lane's own value (with the default width of 64 a 32-lane plane
    // would read the registers of lanes that do not exist)
    case MI355_PLANE_SHUFFLE: out[i] = __shfl(v, (int)arg, (int)plane); break;
    case MI355_PLANE_SHUFFLE_XOR: out[i] = arg >= plane ? v : __shfl_xor(v, (int)arg, (int)plane); break;   // (lane ^ mask outside the plane: own value)
    // (a delta of a whole plane or more has no source lane inside the plane: every lane keeps its own value -- decided here,
    //  because __shfl_up / __shfl_down do their index arithmetic in signed int and a delta >= 2^31 would wrap)
    case MI355_PLANE_SHUFFLE_UP: out[i] = arg >= plane ? v : __shfl_up(v, arg, (int)plane); break;
    case MI355_PLANE_SHUFFLE_DOWN: out[i] = arg >= plane ? v : __shfl_down(v, arg, (int)plane); break;
    case MI355_PLANE_BALLOT: {
        const unsigned long long m = __ballot(v != 0.f);     // 64-bit on this target: words 0 and 1 of the reference's 4 x u32
        if (lane == 0) {
            uint32_t *o = static_cast<uint32_t *>(out_raw) + (uint64_t)blockIdx.x * 4;
            o[0] = (uint32_t)m; o[1] = (uint32_t)(m >> 32); o[2] = 0; o[3] = 0;
        }
        break;
    }
    default: break;
    }
}

// ---- multi-GPU argmax: the combine step -------------------------------------------------------------
// After the all-gather every rank holds one record {f32 value, u32 pad, u64 LOCAL index} per shard (the bytes
// mi355_sum_argmax_f32 wrote at out_val / out_idx, 16 bytes apart).  One wave folds them with the rule the
// single-GPU kernel uses inside a device (arg_combine over argmax_key: larger value wins, NaN above everything,
// -0 == +0, ties keep the LOWEST global index), so the winner lands in device memory on every rank, stream-ordered
// behind the collective -- no host round trip inside the exchange.  An empty shard passes index 2^64-1 and is skipped.
struct combine_bases { uint64_t base[64]; };

// SUMS (round 5): the record's second word carries the shard's partial sum (where mi355_sum_argmax_f32 writes out_sum when it
// is 4 bytes behind out_val), so ONE all-gather moves everything the job exchanges; lane 0 then adds the partial sums
// in RANK ORDER -- a fixed tree, the same bits on every rank and every run (an all-reduce leaves the order to RCCL's
// algorithm choice).  An empty shard's sum word must be +0.0.
template <bool SUMS>
__global__ void __launch_bounds__(64)
argmax_combine_kernel(const uint32_t *__restrict__ records, uint32_t count, combine_bases bases, float *__restrict__ out_sum, float *__restrict__ out_val,
                      uint64_t *__restrict__ out_idx)
{
    const uint32_t lane = threadIdx.x;
    uint32_t key = 0, bits = 0xFF800000u;                   // -inf: what mi355_argmax_f32 reports for an empty array
    uint64_t idx = ~0ull;
    float part = 0.f;
    if (lane < count) {
        const uint32_t vb = records[lane * 4];
        const uint64_t li = ((uint64_t)records[lane * 4 + 3] << 32) | records[lane * 4 + 2];
        if (li != ~0ull) { key = argmax_key(__uint_as_float(vb)); idx = bases.base[lane] + li; bits = vb; }
        if (SUMS) part = __uint_as_float(records[lane * 4 + 1]);
    }
    if (SUMS) {
        float total = 0.f;
        for (uint32_t r = 0; r < count; ++r) total += __shfl(part, (int)r, 64);      // rank order (count is wave-uniform)
        if (lane == 0 && out_sum) *out_sum = total;
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t okey = __shfl_xor(key, off, 64), obits = __shfl_xor(bits, off, 64);
        const uint32_t olo = __shfl_xor((uint32_t)idx, off, 64), ohi = __shfl_xor((uint32_t)(idx >> 32), off, 64);
        const uint64_t oidx = ((uint64_t)ohi << 32) | olo;
        const bool take = (okey > key) || (okey == key && oidx < idx);
        key = take ? okey : key; idx = take ? oidx : idx; bits = take ? obits : bits;
    }
    if (lane == 0) {
        if (out_val) *out_val = __uint_as_float(bits);
        if (out_idx) *out_idx = (idx == ~0ull) ? 0ull : idx;
    }
}

}  // namespace

MI355_API int32_t mi355_reduce_workspace_bytes(mi355_ctx *ctx, uint64_t n, uint64_t *out_bytes)
{
    if (!out_bytes) return MI355_E_INVALID_ARGUMENT;
    (void)ctx; // a constant of the kernels, not of the device: callers without a context (client-side planning) pass NULL
    (void)n;
    *out_bytes = (uint64_t)RED_MAX_GRID * sizeof(red_record) + 256;
    return MI355_OK;
}

MI355_API int32_t mi355_reduce_sum_f32(mi355_ctx *ctx, mi355_stream stream, const float *in, uint64_t n, float *out,
                                       void *workspace, uint64_t workspace_bytes)
{
    if (ctx && !out) return fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_reduce_sum_f32: out is NULL");
    return run_reduce<VOP_SUM, AOP_NONE>(ctx, stream, in, MI355_DTYPE_F32, n, out, nullptr, nullptr, workspace, workspace_bytes,
                                   "mi355_reduce_sum_f32");
}

// ---- the same reductions for f32 / bf16 / f16 inputs (widened to f32 on load: sums and compares in f32) ----------------
MI355_API int32_t mi355_reduce_sum(mi355_ctx *ctx, mi355_stream stream, const void *in, int32_t dtype, uint64_t n, float *out,
                                   void *workspace, uint64_t workspace_bytes)
{
    if (ctx && !out) return fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_reduce_sum: out is NULL");
    return run_reduce<VOP_SUM, AOP_NONE>(ctx, stream, in, dtype, n, out, nullptr, nullptr, workspace, workspace_bytes, "mi355_reduce_sum");
}

MI355_API int32_t mi355_argmax(mi355_ctx *ctx, mi355_stream stream, const void *in, int32_t dtype, uint64_t n, float *out_val,
                               uint64_t *out_idx, void *workspace, uint64_t workspace_bytes)
{
    if (ctx && !out_idx && !out_val) return fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_argmax: no output");
    return run_reduce<VOP_NONE, AOP_MAX>(ctx, stream, in, dtype, n, nullptr, out_val, out_idx, workspace, workspace_bytes, "mi355_argmax");
}

MI355_API int32_t mi355_sum_argmax(mi355_ctx *ctx, mi355_stream stream, const void *in, int32_t dtype, uint64_t n, float *out_sum,
                                   float *out_val, uint64_t *out_idx, void *workspace, uint64_t workspace_bytes)
{
    if (ctx && !out_sum && !out_idx && !out_val) return fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_sum_argmax: no output");
    return run_reduce<VOP_SUM, AOP_MAX>(ctx, stream, in, dtype, n, out_sum, out_val, out_idx, workspace, workspace_bytes, "mi355_sum_argmax");
}

MI355_API int32_t mi355_argmax_f32(mi355_ctx *ctx, mi355_stream stream, const float *in, uint64_t n, float *out_val,
                                   uint64_t *out_idx, void *workspace, uint64_t workspace_bytes)
{
    if (ctx && !out_idx && !out_val) return fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_argmax_f32: no output");
    return run_reduce<VOP_NONE, AOP_MAX>(ctx, stream, in, MI355_DTYPE_F32, n, nullptr, out_val, out_idx, workspace, workspace_bytes,
                                   "mi355_argmax_f32");
}

MI355_API int32_t mi355_sum_argmax_f32(mi355_ctx *ctx, mi355_stream stream, const float *in, uint64_t n,
                                       float *out_sum, float *out_val, uint64_t *out_idx, void *workspace,
                                       uint64_t workspace_bytes)
{
    if (ctx && !out_sum && !out_idx && !out_val)
        return fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_sum_argmax_f32: no output");
    return run_reduce<VOP_SUM, AOP_MAX>(ctx, stream, in, MI355_DTYPE_F32, n, out_sum, out_val, out_idx, workspace, workspace_bytes,
                                  "mi355_sum_argmax_f32");
}

MI355_API int32_t mi355_argmax_combine_f32(mi355_ctx *ctx, mi355_stream stream, const void *records, uint32_t count,
                                           const uint64_t *index_base, float *out_val, uint64_t *out_idx)
{
    MI355_REQUIRE_CTX(ctx);
    if (!out_val && !out_idx) return fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_argmax_combine_f32: no output");
    if (count > 64) return fail(ctx, MI355_E_UNSUPPORTED, "mi355_argmax_combine_f32: %u records (at most 64 shards)", count);
    if (count && (!records || (reinterpret_cast<uintptr_t>(records) & 7u)))
        return fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_argmax_combine_f32: records must be an 8-byte aligned device pointer");
    combine_bases b{};
    for (uint32_t r = 0; r < count; ++r) b.base[r] = index_base ? index_base[r] : 0;
    hipLaunchKernelGGL(argmax_combine_kernel<false>, dim3(1), dim3(64), 0, stream_of(ctx, stream), static_cast<const uint32_t *>(records), count,
                       b, (float *)nullptr, out_val, out_idx);
    check_launch(ctx, "mi355_argmax_combine_f32");
    return MI355_OK;
}

MI355_API int32_t mi355_sum_argmax_combine_f32(mi355_ctx *ctx, mi355_stream stream, const void *records, uint32_t count,
                                               const uint64_t *index_base, float *out_sum, float *out_val, uint64_t *out_idx)
{
    MI355_REQUIRE_CTX(ctx);
    if (!out_sum && !out_val && !out_idx) return fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_sum_argmax_combine_f32: no output");
    if (count > 64) return fail(ctx, MI355_E_UNSUPPORTED, "mi355_sum_argmax_combine_f32: %u records (at most 64 shards)", count);
    if (count && (!records || (reinterpret_cast<uintptr_t>(records) & 7u)))
        return fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_sum_argmax_combine_f32: records must be an 8-byte aligned device pointer");
    combine_bases b{};
    for (uint32_t r = 0; r < count; ++r) b.base[r] = index_base ? index_base[r] : 0;
    hipLaunchKernelGGL(argmax_combine_kernel<true>, dim3(1), dim3(64), 0, stream_of(ctx, stream), static_cast<const uint32_t *>(records), count,
                       b, out_sum, out_val, out_idx);
    check_launch(ctx, "mi355_sum_argmax_combine_f32");
    return MI355_OK;
}

// all-gather + fence + combine of the sharded sum + argmax in one call (header: mi355_sum_argmax_exchange; lives beside the combine
// kernel so that comm.cpp stays free of kernel symbols -- the host-runtime test library links it without reduce.hip)
MI355_API int32_t mi355_sum_argmax_exchange(mi355_ctx *ctx, mi355_comm *comm, mi355_stream stream, const void *record, void *gathered,
                                            const uint64_t *index_base, float *out_sum, float *out_val, uint64_t *out_idx)
{
    MI355_REQUIRE_CTX(ctx);
    if (!comm) return fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_sum_argmax_exchange: communicator is NULL (call mi355_comm_init)");
    if (!record || !gathered) return fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_sum_argmax_exchange: NULL record buffer");
    int32_t rc = mi355_all_gather(ctx, comm, stream, record, gathered, 2, MI355_DTYPE_U64);
    if (rc == MI355_OK) rc = mi355_sync_collective(ctx, stream);
    if (rc == MI355_OK) rc = mi355_sum_argmax_combine_f32(ctx, stream, gathered, (uint32_t)comm_world_size(comm), index_base, out_sum, out_val, out_idx);
    return rc;
}

MI355_API int32_t mi355_reduce_last_axis_sum_f32(mi355_ctx *ctx, mi355_stream stream, const float *in, float *out,
                                                 uint64_t rows, uint64_t cols, uint64_t row_stride)
{
    return run_rows<MI355_REDUCE_SUM>(ctx, stream, in, out, nullptr, rows, cols, row_stride, "mi355_reduce_last_axis_sum_f32");
}

MI355_API int32_t mi355_reduce_last_axis_argmax_f32(mi355_ctx *ctx, mi355_stream stream, const float *in,
                                                    uint32_t *out_idx, uint64_t rows, uint64_t cols,
                                                    uint64_t row_stride)
{
    return run_rows<MI355_REDUCE_ARGMAX>(ctx, stream, in, nullptr, out_idx, rows, cols, row_stride,
                          "mi355_reduce_last_axis_argmax_f32");
}

// last-axis reductions of f32 / bf16 / f16 rows (f32 arithmetic, f32 sums / u32 indices out)
MI355_API int32_t mi355_reduce_last_axis_sum(mi355_ctx *ctx, mi355_stream stream, const void *in, int32_t dtype, float *out,
                                             uint64_t rows, uint64_t cols, uint64_t row_stride)
{
    if (!ctx) return MI355_E_INVALID_ARGUMENT;
    switch (dtype) {
    case MI355_DTYPE_F32: return run_rows<MI355_REDUCE_SUM>(ctx, stream, static_cast<const float *>(in), out, nullptr, rows, cols, row_stride, "mi355_reduce_last_axis_sum");
    case MI355_DTYPE_BF16: return run_rows<MI355_REDUCE_SUM, MI355_DTYPE_BF16>(ctx, stream, static_cast<const uint16_t *>(in), out, nullptr, rows, cols, row_stride, "mi355_reduce_last_axis_sum");
    case MI355_DTYPE_F16: return run_rows<MI355_REDUCE_SUM, MI355_DTYPE_F16>(ctx, stream, static_cast<const uint16_t *>(in), out, nullptr, rows, cols, row_stride, "mi355_reduce_last_axis_sum");
    default: return fail(ctx, MI355_E_UNSUPPORTED, "mi355_reduce_last_axis_sum: input dtype %d (f32, bf16 or f16)", dtype);
    }
}

MI355_API int32_t mi355_reduce_last_axis_argmax(mi355_ctx *ctx, mi355_stream stream, const void *in, int32_t dtype, uint32_t *out_idx,
                                                uint64_t rows, uint64_t cols, uint64_t row_stride)
{
    if (!ctx) return MI355_E_INVALID_ARGUMENT;
    switch (dtype) {
    case MI355_DTYPE_F32: return run_rows<MI355_REDUCE_ARGMAX>(ctx, stream, static_cast<const float *>(in), nullptr, out_idx, rows, cols, row_stride, "mi355_reduce_last_axis_argmax");
    case MI355_DTYPE_BF16: return run_rows<MI355_REDUCE_ARGMAX, MI355_DTYPE_BF16>(ctx, stream, static_cast<const uint16_t *>(in), nullptr, out_idx, rows, cols, row_stride, "mi355_reduce_last_axis_argmax");
    case MI355_DTYPE_F16: return run_rows<MI355_REDUCE_ARGMAX, MI355_DTYPE_F16>(ctx, stream, static_cast<const uint16_t *>(in), nullptr, out_idx, rows, cols, row_stride, "mi355_reduce_last_axis_argmax");
    default: return fail(ctx, MI355_E_UNSUPPORTED, "mi355_reduce_last_axis_argmax: input dtype %d (f32, bf16 or f16)", dtype);
    }
}

MI355_API int32_t mi355_reduce_axis_sum_f32(mi355_ctx *ctx, mi355_stream stream, const float *in, float *out, uint64_t outer,
                                            uint64_t reduce, uint64_t inner)
{
    return mi355_reduce_axis_sum(ctx, stream, in, MI355_DTYPE_F32, out, outer, reduce, inner);
}

MI355_API int32_t mi355_reduce_axis_argmax_f32(mi355_ctx *ctx, mi355_stream stream, const float *in, uint32_t *out_idx,
                                               uint64_t outer, uint64_t reduce, uint64_t inner)
{
    return mi355_reduce_axis_argmax(ctx, stream, in, MI355_DTYPE_F32, out_idx, outer, reduce, inner);
}

// any-axis reductions of f32 / bf16 / f16 input (f32 arithmetic)
template <int OP, int DT>
static int32_t axis_dispatch(mi355_ctx *ctx, mi355_stream stream, const void *in, float *out, uint32_t *out_idx, uint64_t outer,
                             uint64_t reduce, uint64_t inner, const char *what)
{
    typedef typename red_in<DT>::elem elem;
    // last axis: a wave (or a workgroup) per row -- unless the rows are so short that most of its lanes would idle (16 Mi x 4: 140 GB/s,
    // 4 Mi x 16: 550, 1 Mi x 64: 2 000 -- profiles/r04_misc_probe.txt); the tiled kernel then packs 256 / TY rows into a workgroup
    if (inner == 1 && reduce >= 128) return run_rows<OP, DT>(ctx, stream, static_cast<const elem *>(in), out, out_idx, outer, reduce, reduce, what);
    return run_mid<OP, DT>(ctx, stream, static_cast<const elem *>(in), out, out_idx, outer, reduce, inner, what);
}
template <int OP>
static int32_t axis_dispatch_dtype(mi355_ctx *ctx, mi355_stream stream, const void *in, int32_t dtype, float *out, uint32_t *out_idx,
                                   uint64_t outer, uint64_t reduce, uint64_t inner, const char *what)
{
    switch (dtype) {
    case MI355_DTYPE_F32: return axis_dispatch<OP, MI355_DTYPE_F32>(ctx, stream, in, out, out_idx, outer, reduce, inner, what);
    case MI355_DTYPE_BF16: return axis_dispatch<OP, MI355_DTYPE_BF16>(ctx, stream, in, out, out_idx, outer, reduce, inner, what);
    case MI355_DTYPE_F16: return axis_dispatch<OP, MI355_DTYPE_F16>(ctx, stream, in, out, out_idx, outer, reduce, inner, what);
    default: return fail(ctx, MI355_E_UNSUPPORTED, "%s: input dtype %d (f32, bf16 or f16)", what, dtype);
    }
}

MI355_API int32_t mi355_reduce_axis_sum(mi355_ctx *ctx, mi355_stream stream, const void *in, int32_t dtype, float *out,
                                        uint64_t outer, uint64_t reduce, uint64_t inner)
{
    if (!ctx) return MI355_E_INVALID_ARGUMENT;
    switch (dtype) {
    case MI355_DTYPE_F32: return axis_dispatch<MI355_REDUCE_SUM, MI355_DTYPE_F32>(ctx, stream, in, out, nullptr, outer, reduce, inner, "mi355_reduce_axis_sum");
    case MI355_DTYPE_BF16: return axis_dispatch<MI355_REDUCE_SUM, MI355_DTYPE_BF16>(ctx, stream, in, out, nullptr, outer, reduce, inner, "mi355_reduce_axis_sum");
    case MI355_DTYPE_F16: return axis_dispatch<MI355_REDUCE_SUM, MI355_DTYPE_F16>(ctx, stream, in, out, nullptr, outer, reduce, inner, "mi355_reduce_axis_sum");
    default: return fail(ctx, MI355_E_UNSUPPORTED, "mi355_reduce_axis_sum: input dtype %d (f32, bf16 or f16)", dtype);
    }
}

MI355_API int32_t mi355_reduce_axis_argmax(mi355_ctx *ctx, mi355_stream stream, const void *in, int32_t dtype, uint32_t *out_idx,
                                           uint64_t outer, uint64_t reduce, uint64_t inner)
{
    if (!ctx) return MI355_E_INVALID_ARGUMENT;
    switch (dtype) {
    case MI355_DTYPE_F32: return axis_dispatch<MI355_REDUCE_ARGMAX, MI355_DTYPE_F32>(ctx, stream, in, nullptr, out_idx, outer, reduce, inner, "mi355_reduce_axis_argmax");
    case MI355_DTYPE_BF16: return axis_dispatch<MI355_REDUCE_ARGMAX, MI355_DTYPE_BF16>(ctx, stream, in, nullptr, out_idx, outer, reduce, inner, "mi355_reduce_axis_argmax");
    case MI355_DTYPE_F16: return axis_dispatch<MI355_REDUCE_ARGMAX, MI355_DTYPE_F16>(ctx, stream, in, nullptr, out_idx, outer, reduce, inner, "mi355_reduce_axis_argmax");
    default: return fail(ctx, MI355_E_UNSUPPORTED, "mi355_reduce_axis_argmax: input dtype %d (f32, bf16 or f16)", dtype);
    }
}

// ---- every reduce operation through one pair of entry points per form (round 4) ----------------------------------------------
MI355_API int32_t mi355_reduce(mi355_ctx *ctx, mi355_stream stream, const void *in, int32_t dtype, uint64_t n, int32_t op, float *out,
                               void *workspace, uint64_t workspace_bytes)
{
    if (ctx && !out) return fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_reduce: out is NULL");
    switch (op) {
    case MI355_REDUCE_SUM: return run_reduce<VOP_SUM, AOP_NONE>(ctx, stream, in, dtype, n, out, nullptr, nullptr, workspace, workspace_bytes, "mi355_reduce(sum)");
    case MI355_REDUCE_MEAN: return run_reduce<VOP_SUM, AOP_NONE>(ctx, stream, in, dtype, n, out, nullptr, nullptr, workspace, workspace_bytes, "mi355_reduce(mean)",
                                                                 n ? (float)n : 1.f);
    case MI355_REDUCE_MAX: return run_reduce<VOP_MAX, AOP_NONE>(ctx, stream, in, dtype, n, out, nullptr, nullptr, workspace, workspace_bytes, "mi355_reduce(max)");
    case MI355_REDUCE_MIN: return run_reduce<VOP_MIN, AOP_NONE>(ctx, stream, in, dtype, n, out, nullptr, nullptr, workspace, workspace_bytes, "mi355_reduce(min)");
    case MI355_REDUCE_PROD: return run_reduce<VOP_PROD, AOP_NONE>(ctx, stream, in, dtype, n, out, nullptr, nullptr, workspace, workspace_bytes, "mi355_reduce(prod)");
    default: return ctx ? fail(ctx, MI355_E_UNSUPPORTED, "mi355_reduce: op %d is not a value reduction (SUM, MEAN, MAX, MIN, PROD)", op) : MI355_E_INVALID_ARGUMENT;
    }
}

MI355_API int32_t mi355_argreduce(mi355_ctx *ctx, mi355_stream stream, const void *in, int32_t dtype, uint64_t n, int32_t op, float *out_val,
                                  uint64_t *out_idx, void *workspace, uint64_t workspace_bytes)
{
    if (ctx && !out_idx && !out_val) return fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_argreduce: no output");
    switch (op) {
    case MI355_REDUCE_ARGMAX: return run_reduce<VOP_NONE, AOP_MAX>(ctx, stream, in, dtype, n, nullptr, out_val, out_idx, workspace, workspace_bytes, "mi355_argreduce(argmax)");
    case MI355_REDUCE_ARGMIN: return run_reduce<VOP_NONE, AOP_MIN>(ctx, stream, in, dtype, n, nullptr, out_val, out_idx, workspace, workspace_bytes, "mi355_argreduce(argmin)");
    default: return ctx ? fail(ctx, MI355_E_UNSUPPORTED, "mi355_argreduce: op %d is not an index reduction (ARGMAX, ARGMIN)", op) : MI355_E_INVALID_ARGUMENT;
    }
}

MI355_API int32_t mi355_reduce_axis(mi355_ctx *ctx, mi355_stream stream, const void *in, int32_t dtype, int32_t op, float *out,
                                    uint64_t outer, uint64_t reduce, uint64_t inner)
{
    if (!ctx) return MI355_E_INVALID_ARGUMENT;
    switch (op) {
    case MI355_REDUCE_SUM: return axis_dispatch_dtype<MI355_REDUCE_SUM>(ctx, stream, in, dtype, out, nullptr, outer, reduce, inner, "mi355_reduce_axis(sum)");
    case MI355_REDUCE_MEAN: return axis_dispatch_dtype<MI355_REDUCE_MEAN>(ctx, stream, in, dtype, out, nullptr, outer, reduce, inner, "mi355_reduce_axis(mean)");
    case MI355_REDUCE_MAX: return axis_dispatch_dtype<MI355_REDUCE_MAX>(ctx, stream, in, dtype, out, nullptr, outer, reduce, inner, "mi355_reduce_axis(max)");
    case MI355_REDUCE_MIN: return axis_dispatch_dtype<MI355_REDUCE_MIN>(ctx, stream, in, dtype, out, nullptr, outer, reduce, inner, "mi355_reduce_axis(min)");
    case MI355_REDUCE_PROD: return axis_dispatch_dtype<MI355_REDUCE_PROD>(ctx, stream, in, dtype, out, nullptr, outer, reduce, inner, "mi355_reduce_axis(prod)");
    default: return fail(ctx, MI355_E_UNSUPPORTED, "mi355_reduce_axis: op %d is not a value reduction (SUM, MEAN, MAX, MIN, PROD)", op);
    }
}

MI355_API int32_t mi355_argreduce_axis(mi355_ctx *ctx, mi355_stream stream, const void *in, int32_t dtype, int32_t op, uint32_t *out_idx,
                                       uint64_t outer, uint64_t reduce, uint64_t inner)
{
    if (!ctx) return MI355_E_INVALID_ARGUMENT;
    switch (op) {
    case MI355_REDUCE_ARGMAX: return axis_dispatch_dtype<MI355_REDUCE_ARGMAX>(ctx, stream, in, dtype, nullptr, out_idx, outer, reduce, inner, "mi355_argreduce_axis(argmax)");
    case MI355_REDUCE_ARGMIN: return axis_dispatch_dtype<MI355_REDUCE_ARGMIN>(ctx, stream, in, dtype, nullptr, out_idx, outer, reduce, inner, "mi355_argreduce_axis(argmin)");
    default: return fail(ctx, MI355_E_UNSUPPORTED, "mi355_argreduce_axis: op %d is not an index reduction (ARGMAX, ARGMIN)", op);
    }
}

MI355_API int32_t mi355_plane_reduce_f32(mi355_ctx *ctx, mi355_stream stream, const float *in, float *out, uint64_t n,
                                         uint32_t active, int32_t op)
{
    MI355_REQUIRE_CTX(ctx);
    if (n == 0) return MI355_OK;
    if (!in || !out) return fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_plane_reduce_f32: NULL pointer");
    if (active == 0 || active > 64 || (active & (active - 1)) != 0)
        return fail(ctx, MI355_E_INVALID_ARGUMENT, "active lanes must be a power of two <= 64 (got %u)", active);
    const bool known = op == MI355_REDUCE_SUM || op == MI355_REDUCE_MAX || op == MI355_REDUCE_MIN || op == MI355_REDUCE_PROD || op == 100 ||
                       (op >= MI355_PLANE_INCLUSIVE_SUM && op <= MI355_PLANE_EXCLUSIVE_PROD);
    if (!known) return fail(ctx, MI355_E_UNSUPPORTED, "unknown plane op %d", op);
    const uint64_t blocks = std::min<uint64_t>(((n + 63) / 64 + 3) / 4, (uint64_t)ctx->props.num_streaming_multiprocessors * 32);
    hipLaunchKernelGGL(plane_reduce_kernel, dim3((uint32_t)std::max<uint64_t>(blocks, 1)), dim3(256), 0, stream_of(ctx, stream), in, out, n,
                       active, op);
    check_launch(ctx, "mi355_plane_reduce_f32");
    return MI355_OK;
}

MI355_API int32_t mi355_plane_op_f32(mi355_ctx *ctx, mi355_stream stream, const float *in, void *out, uint64_t n, uint32_t plane,
                                     int32_t op, uint32_t arg)
{
    MI355_REQUIRE_CTX(ctx);
    if (n == 0) return MI355_OK;
    if (!in || !out) return fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_plane_op_f32: NULL pointer");
    if (plane != 32 && plane != 64) return fail(ctx, MI355_E_INVALID_ARGUMENT, "plane must be 32 or 64 lanes (got %u)", plane);
    if (op < MI355_PLANE_ALL || op > MI355_PLANE_BALLOT) return fail(ctx, MI355_E_UNSUPPORTED, "unknown plane op %d", op);
    if ((op == MI355_PLANE_BROADCAST || op == MI355_PLANE_SHUFFLE) && arg >= plane)
        return fail(ctx, MI355_E_INVALID_ARGUMENT, "source lane %u outside the plane of %u", arg, plane);
    const uint64_t blocks = (n + plane - 1) / plane;
    if (blocks > 0x7FFFFFFFull) return fail(ctx, MI355_E_UNSUPPORTED, "too many planes");
    hipLaunchKernelGGL(plane_op_kernel, dim3((uint32_t)blocks), dim3(plane), 0, stream_of(ctx, stream), in, out, n, (int)op, arg);
    check_launch(ctx, "mi355_plane_op_f32");
    return MI355_OK;
}
